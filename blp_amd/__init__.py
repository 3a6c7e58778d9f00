"""blp_amd -- MI355X-native link-prediction scoring / ranking hot path of dfdazac/blp.

    blp_amd.models   LinkPrediction interface (reference models.py), HIP-backed on a GPU
    blp_amd.ranking  all-entities ranking evaluation, CSR filters, candidate-axis sharding
    blp_amd.ops      torch-facing operators over the C-ABI (include/blp_hip.h, libblp_hip.so)
    blp_amd.utils / blp_amd.data   host-side helpers with the reference's names
    blp_amd.build    builds libblp_hip.so with hipcc (python -m blp_amd.build)
"""
__version__ = "0.1.0"
