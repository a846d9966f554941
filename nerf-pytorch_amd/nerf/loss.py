"""PSNR from MSE (reference nerf/loss.py): 20*log10(1/sqrt(mse)), host-side scalar math."""
import math


def mse_to_psnr(mse: float) -> float:
    return 20.0 * math.log10(1.0 / math.sqrt(mse))
