"""nicer_slam_amd -- MI355X-native (gfx950) per-frame neural rendering core for NICER-SLAM.

Only what the hot path needs: ``csrc/`` (HIP kernels + the C ABI declared in ``include/nicer_slam_amd.h``),
``hashencoder/`` (drop-in for the reference's ``code/hashencoder`` package) and ``model/`` (drop-in for
``code/model``: ``train.model_class = nicer_slam_amd.model.network.SLAMNetwork``).
"""
__version__ = "0.1.0"
