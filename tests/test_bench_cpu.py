"""bench.py host logic without a GPU: the workload label follows the arguments actually run, the BASELINE presets expand."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ns(**kw):
    d = dict(rays=1024, samples=128, precision="fp32", global_rays=0, config=0)
    d.update(kw)
    return argparse.Namespace(**d)


def test_workload_label_follows_the_arguments():
    sys.path.insert(0, ROOT)
    import bench
    one = bench.workload_label(_ns(), 1, False)
    assert "1024 rays x 128 samples" in one and "single MI355X" in one and "[BASELINE configs[1]]" in one and "fp32" in one
    two = bench.workload_label(_ns(), 2, False)
    assert "2048 rays" in two and "2 ranks ray-sharded (1024 rays/rank, weak scaling)" in two and "RCCL" in two
    assert "single MI355X" not in two and "configs[1] per GPU x 2" in two
    c2 = bench.workload_label(_ns(rays=512, global_rays=4096, precision="bf16"), 8, False)
    assert "4096 rays x 128" in c2 and "strong scaling" in c2 and "bf16 MLP operands" in c2 and "[BASELINE configs[2]]" in c2
    c4 = bench.workload_label(_ns(rays=1024, global_rays=8192, samples=192, precision="bf16_colour"), 8, False)
    assert "8192 rays x 192" in c4 and "fp32 SDF head" in c4 and "configs[4]" in c4
    smoke = bench.workload_label(_ns(), 2, True)
    assert "OVERSUBSCRIBED" in smoke and "RCCL" not in smoke
    odd = bench.workload_label(_ns(rays=300, samples=64), 1, False)
    assert "no BASELINE config" in odd


def test_config_presets_expand_and_reject_contradictions():
    run = lambda *a: subprocess.run([sys.executable, "-c",
                                     "import sys; sys.argv=['bench.py']+%r; import bench; a=bench.parse(); "
                                     "print(a.gpus, a.global_rays, a.samples, a.precision)" % (list(a),)],
                                    capture_output=True, text=True, cwd=ROOT)
    assert run("--config", "2").stdout.split() == ["8", "4096", "128", "bf16"]
    assert run("--config", "4", "--gpus", "8").stdout.split() == ["8", "8192", "192", "bf16_colour"]
    assert run("--config", "1").stdout.split() == ["1", "0", "128", "fp32"]
    bad = run("--config", "2", "--gpus", "4")
    assert bad.returncode != 0 and "means --gpus 8" in bad.stderr


def test_mapping_leg_quotes_committed_profiles_consistently():
    """The mapping leg's context numbers read committed files: the in-library share comes from the newest rNN_mapping_kernel_stats.csv
    (normalised by the once-per-iteration kernel, whatever number of iterations the profile held), and the atomic-request counters are
    only quoted when their signature file matches the batch shape, tilings and key bits of the run."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    share = bench.mapping_library_share()
    assert share is not None and share["source"].startswith("profiles/r") and 0.9 < share["share"] <= 1.0
    assert 100 < share["launches_per_iteration"] < 400
    sig = bench.mapping_signature()
    assert sig["rays"] == 8192 and sig["morton_bits"] in range(1, 31) and "fine" in sig["tiles"]
    metas = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_mapping_pmc_meta.json"))
    newest = json.load(open(os.path.join(ROOT, "profiles", metas[-1])))
    assert newest == sig, "the newest committed atomic-request profile was taken with another shape / tiling / key width"
