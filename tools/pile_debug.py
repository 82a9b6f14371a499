import numpy as np, sys
sys.path.insert(0,'/root/repo')
from tests.oracle_api import Oracle
from moby_amd import scene as S
from moby_amd.world import WorldBatch
from moby_amd.synth import world_uniforms
o=Oracle('/root/repo/oracle/liboracle.so')
radii=[0.5]*8; cp = dict(epsilon=0.1, mu_coulomb=0.3, mu_viscous=0.0, nk=4)
params = {(i, j): cp for i in range(8) for j in range(i + 1, 9)}
sc = S.make_scene(radii, [1.0] * 8, (0.0, -9.81, 0.0), ground_rpy=(0.0, 0.0, 0.0), params=params); sc.cstab_max_iterations=10
sts=[]
for w in range(4):
    u = world_uniforms(w, 24)
    st = np.zeros((8, 13)); st[:, 6] = 1.0
    for b in range(8):
        st[b, :3] = (1.05 * (b % 3) + 0.1 * u[b], 0.5 + 1.02 * (b // 3) + 0.05 * u[8 + b], 0.3 * (u[16 + b] - 0.5))
    sts.append(st.ravel())
st0=np.array(sts)
wb=WorldBatch(sc,st0.copy()); so=st0.copy(); ao=S.new_aux(4)
for k in range(400):
    wb.step(1e-3,1)
    for w in range(4): o.world_step(sc,so[w],ao[w:w+1],1e-3,1,want_traj=False)
    bad=[w for w in range(4) if not np.array_equal(wb.state[w],so[w])]
    if bad or (wb.aux['status']!=ao['status']).any():
        print("step",k,"bad worlds",bad,"gpu status",wb.aux['status'],"oracle",ao['status'])
        for w in bad[:1]:
            print(" gpu lcp",wb.aux['lcp_solves'][w],wb.aux['lcp_rows'][w],wb.aux['lcp_pivots'][w],"oracle",ao['lcp_solves'][w],ao['lcp_rows'][w],ao['lcp_pivots'][w], "mini", wb.aux['mini_steps'][w], ao['mini_steps'][w], "stab", wb.aux['stab_iters'][w], ao['stab_iters'][w])
            d=np.abs(wb.state[w]-so[w]).reshape(8,13); print(" max diff per body",d.max(axis=1))
        break
else: print("all equal")
