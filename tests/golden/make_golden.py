"""Generates the committed golden fixtures from the reference's own regression
data (run in the build container, where /root/reference exists):

  sphere_stack_dat.npz  rows of /root/reference/regress/sphere-stack.dat
                        (t + 3 x [x y z qx qy qz qw], printed with 6 significant
                        digits by programs/regress.cpp:82-93): the first 30 rows,
                        then every 25th, and the last row; plus the CPU-seconds
                        footer (regress.cpp:274-277).

  rimless_wheel_dat.npz rows of /root/reference/regress/rimless-wheel.dat (t + the wheel's
                        [x y z qx qy qz qw]): the first 30 rows, then every 25th, and the
                        last row (the file ends at t = 6.274 s without a footer).

  sitting_box_dat.npz   rows of /root/reference/regress/sitting-box.dat (t + the box's
                        [x y z qx qy qz qw]; 10001 rows, dt = 1e-3): first 30, every 100th, last;
                        plus the CPU-seconds footer.

  contact_constrained_pendulum_dat.npz
                        ALL rows of /root/reference/regress/contact-constrained-pendulum.dat (t + the link's
                        [x y z qx qy qz qw]; 6500 rows, dt = 1e-3, 6.5 s of a swinging pendulum): the one DYNAMIC
                        trajectory among the reference's regression files that a scene in scope produces.

Only DATA is stored (inputs / expected outputs), never reference source.
"""
import os
import numpy as np

REF = "/root/reference/regress"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    lines = open(os.path.join(REF, "sphere-stack.dat")).read().split("\n")
    rows = [l for l in lines if len(l.split()) == 22]
    footer = [l for l in lines if len(l.split()) == 1 and l.strip()]
    data = np.array([[float(x) for x in r.split()] for r in rows])
    keep = sorted(set(list(range(30)) + list(range(0, len(data), 25)) + [len(data) - 1]))
    np.savez_compressed(os.path.join(HERE, "sphere_stack_dat.npz"), row_index=np.array(keep), rows=data[keep],
                        n_rows=len(data), cpu_seconds=float(footer[-1]) if footer else np.nan)
    print("sphere-stack.dat: %d rows -> %d kept" % (len(data), len(keep)))

    lines = open(os.path.join(REF, "rimless-wheel.dat")).read().split("\n")
    data = np.array([[float(x) for x in l.split()] for l in lines if len(l.split()) == 8])
    keep = sorted(set(list(range(30)) + list(range(0, len(data), 25)) + [len(data) - 1]))
    np.savez_compressed(os.path.join(HERE, "rimless_wheel_dat.npz"), row_index=np.array(keep), rows=data[keep], n_rows=len(data))
    print("rimless-wheel.dat: %d rows -> %d kept" % (len(data), len(keep)))

    lines = open(os.path.join(REF, "sitting-box.dat")).read().split("\n")
    data = np.array([[float(x) for x in l.split()] for l in lines if len(l.split()) == 8])
    footer = [l for l in lines if len(l.split()) == 1 and l.strip()]
    keep = sorted(set(list(range(30)) + list(range(0, len(data), 100)) + [len(data) - 1]))
    np.savez_compressed(os.path.join(HERE, "sitting_box_dat.npz"), row_index=np.array(keep), rows=data[keep], n_rows=len(data),
                        cpu_seconds=float(footer[-1]) if footer else np.nan)
    print("sitting-box.dat: %d rows -> %d kept" % (len(data), len(keep)))

    lines = open(os.path.join(REF, "contact-constrained-pendulum.dat")).read().split("\n")
    data = np.array([[float(x) for x in l.split()] for l in lines if len(l.split()) == 8])
    footer = [l for l in lines if len(l.split()) == 1 and l.strip()]
    np.savez_compressed(os.path.join(HERE, "contact_constrained_pendulum_dat.npz"), rows=data, n_rows=len(data),
                        cpu_seconds=float(footer[-1]) if footer else np.nan)
    print("contact-constrained-pendulum.dat: %d rows (all kept)" % len(data))


if __name__ == "__main__":
    main()
