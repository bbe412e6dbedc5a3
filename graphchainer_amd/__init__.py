"""MI355X-native GraphChainer hot path: minimizer seeding, fragment seed extension, co-linear chaining.

The compute path is the HIP library graphchainer_amd/libgraphchainer_amd.so (C ABI: include/graphchainer_amd.h);
this package is its Python host side. Importing the package does not load the library; using it does, and fails
loudly when the library or a GPU is missing.
"""
from . import api  # noqa: F401
from .api import GAM_DEVICE_HUFFMAN, GAM_DEVICE_LZ, Aligner, AlignmentGraph, MinimizerSeeder, ReadBatch, device_count, device_memory, edit_distance, gzip_streams, load_library, set_device, std_sort_permutations  # noqa: F401
