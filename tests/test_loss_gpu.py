"""Row f1 on the device: SLAMLoss (nicer_slam_amd/model/loss.py) with every tensor on the GPU against the goldens captured from
the reference's SLAMLoss (code/model/loss.py:113-233) -- every returned term and the gradient w.r.t. every model output."""
import pytest

from test_loss_cpu import check_slam_loss

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["loss_mapping_first_frame", "loss_mapping_fine"])
def test_slam_loss_on_device_vs_reference_golden(name):
    check_slam_loss(name, device="cuda", atol=2e-6, gtol=2e-7)
