"""Drop-in for the reference's ``code/model`` package: select with
``train.model_class = nicer_slam_amd.model.network.SLAMNetwork``."""
