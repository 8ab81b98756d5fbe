"""Build-container script (needs /root/reference): the first 64 poses of the reference's ground-truth trajectories of Replica room0,
7-Scenes office and Azure 2 (`gt_trajs/gt_replica_room0.txt`, `gt_trajs/gt_7scenes_office.txt`, `gt_trajs/gt_azure_2.txt`: one line per frame, `stamp tx ty tz qx qy qz qw`, what
code/evaluation/eval_cam.py:456-459 compares an estimate against) -> tests/golden/replica_room0_traj64.txt, scenes7_office_traj64.txt,
azure_2_traj64.txt (BASELINE configs[4] names the self-captured azure_2 sequence).
Data fixtures (numbers), used by tools/synthetic_sequence.py as the camera paths of the synthetic multi-frame runs ("poses from
gt_trajs/gt_replica_room0.txt recentred / scaled into the unit cube"; BASELINE configs[3] names the 7-Scenes office sequence)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PAIRS = (("/root/reference/gt_trajs/gt_replica_room0.txt", "replica_room0_traj64.txt"),
         ("/root/reference/gt_trajs/gt_7scenes_office.txt", "scenes7_office_traj64.txt"),
         ("/root/reference/gt_trajs/gt_azure_2.txt", "azure_2_traj64.txt"))

if __name__ == "__main__":
    for src, name in PAIRS:
        rows = [ln for ln in open(src).read().splitlines() if ln.strip()][:64]
        assert len(rows) == 64 and all(len(r.split()) == 8 for r in rows)
        open(os.path.join(HERE, name), "w").write("\n".join(rows) + "\n")
        print(name, len(rows))
