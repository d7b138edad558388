from vidu4d_amd.gs.loss_utils import l1_loss, l2_loss, ssim  # noqa: F401
