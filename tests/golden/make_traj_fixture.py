"""Build-container script (needs /root/reference): the first 64 poses of the reference's ground-truth trajectory of Replica room0
(`gt_trajs/gt_replica_room0.txt`: one line per frame, `stamp tx ty tz qx qy qz qw`, what code/evaluation/eval_cam.py:456-459 compares
an estimate against) -> tests/golden/replica_room0_traj64.txt.  A data fixture (numbers), used by tools/synthetic_sequence.py as the
camera path of the synthetic multi-frame tracking run ("poses from gt_trajs/gt_replica_room0.txt recentred / scaled
into the unit cube")."""
import os

SRC = "/root/reference/gt_trajs/gt_replica_room0.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "replica_room0_traj64.txt")

if __name__ == "__main__":
    rows = [ln for ln in open(SRC).read().splitlines() if ln.strip()][:64]
    assert len(rows) == 64 and all(len(r.split()) == 8 for r in rows)
    open(DST, "w").write("\n".join(rows) + "\n")
    print(DST, len(rows))
