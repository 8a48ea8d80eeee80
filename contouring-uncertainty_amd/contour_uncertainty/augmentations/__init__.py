"""On-device data augmentation with un-apply (test-time augmentation): the reference's ``contour_uncertainty/augmentations``
(CPU, per item, torchvision) re-hosted on whole device batches -- see ``augmentation.py``."""
from contour_uncertainty.augmentations.affine import RandomRotation, RandomTranslation  # noqa: F401
from contour_uncertainty.augmentations.augmentation import Augmentation, Compose, to_tuple  # noqa: F401
from contour_uncertainty.augmentations.brightnesscontrast import RandomBrightnessContrast  # noqa: F401
from contour_uncertainty.augmentations.gamma import RandomGamma  # noqa: F401
