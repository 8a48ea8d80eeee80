"""Synthetic contour data: stands in for the CAMUS HDF5 files, which are not
redistributable, behind the same batch contract as ``CamusContour`` (reference data/camus/dataset.py:100-149)."""
from contour_uncertainty.data.synthetic.datamodule import SyntheticContourDataModule, synthetic_batch  # noqa: F401
