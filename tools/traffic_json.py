#!/usr/bin/env python
"""pmc_summary CSV -> {bench kernel label: HBM bytes per launch} (FETCH_SIZE + WRITE_SIZE, reported in KiB units by
rocprofv3 on gfx950 -> bytes; narrow gathers, so the guide's x2 wide-stream correction is not applied)."""
import csv
import json
import sys

LABEL = [("k_sdfnet4_fwd_pair", "k_sdfnet_fwd<pair>"), ("k_colour_coarse_bwd", "k_colour_coarse_bwd"), ("k_composite_track", "k_composite_track"), ("k_sdfnet4_bwd<8, 4, 3", "k_sdfnet_bwd<fine>"), ("k_sdfnet4_bwd<4, 8, 1", "k_sdfnet_bwd<coarse>"),
         ("k_sdfnet4_fwd<8, 4, 3", "k_sdfnet_fwd<fine>"), ("k_sdfnet4_fwd<4, 8, 1", "k_sdfnet_fwd<coarse>"),
         ("k_sampler4_sdf", "k_sampler_sdf"), ("k_sdfnet_bwd<8, 4, 3", "k_sdfnet_bwd<fine>"), ("k_sdfnet_bwd<4, 8, 1", "k_sdfnet_bwd<coarse>"),
         ("k_sdfnet_fwd<8, 4, 3", "k_sdfnet_fwd<fine>"), ("k_sdfnet_fwd<4, 8, 1", "k_sdfnet_fwd<coarse>"),
         ("k_colour_bwd", "k_colour_bwd"), ("k_colour_fwd", "k_colour_fwd"), ("k_composite_bwd", "k_composite_bwd"),
         ("k_composite_fwd", "k_composite_fwd"), ("k_sample_rays", "k_sample_rays"), ("k_sampler_sdf", "k_sampler_sdf")]
acc = {}
for row in csv.DictReader(open(sys.argv[1])):
    if row["counter"] not in ("FETCH_SIZE", "WRITE_SIZE"):
        continue
    for key, label in LABEL:
        if key in row["kernel"]:
            acc[label] = acc.get(label, 0.0) + float(row["mean_per_dispatch"]) * 1024.0
            break
print(json.dumps({k: int(v) for k, v in sorted(acc.items())}, indent=1))
