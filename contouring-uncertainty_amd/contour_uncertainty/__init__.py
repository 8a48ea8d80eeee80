"""MI355X-native drop-in for the DSNT contour-regression path of ThierryJudge/contouring-uncertainty.

The dotted paths under this package mirror the reference's Hydra ``_target_``s
(``contour_uncertainty.task.regression.dsnt.dsnt_al.DSNTAleatoric``, ``...dsnt_skew.DSNTSkew``,
``contour_uncertainty.models.nnUnet.unet2.UNet``) so that the reference's ``runner.py`` instantiates these classes
unchanged.  All arithmetic runs in libcontour_hip.so (hand-written gfx950 kernels); there is no PyTorch fallback.
"""
