"""MI355X-native field-transform engine for the filter hot path of anemoi-transform.

Import name: ``anemoi_transform_amd`` (the directory is ``anemoi-transform_amd``;
``setup.py`` maps it, and in-tree users call ``__graft_entry__.load_package()``).

Layers:
  native    ctypes binding of libatx.so — hand-written gfx950 HIP kernels (csrc/)
  stack     HBM-resident batches of same-grid fields (column layout)
  grids     formula grids (octahedral / Gaussian / lat-lon)
  interp    CPU precompute of gather indices and weights (cKDTree, bilinear)
  fields    earthkit-shaped Field / FieldList backed by stacks
  core      Transform / Filter / SingleFieldFilter / Pipeline / Registry
  filters   the registered filters: regrid, apply_mask, remove_nans, rescale, ...
"""

__version__ = "0.1.0"
