"""mfs_amd -- MI355X-native moment-filter hot path behind the reference's `moment_filter_*` API.

`mfs_amd.one_dim.filtering` / `mfs_amd.multi_dims.filtering` mirror `mfs.one_dim.filtering` /
`mfs.multi_dims.filtering`; the compute is hand-written HIP (mfs_amd/csrc) behind the C ABI in include/mfs_hip.h.
"""
__version__ = '0.1.0'
