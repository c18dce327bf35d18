"""v2xgnn -- MI355X-native GNN message-passing engine behind the reference `BS` call surface.

Replaces the Keras/TF1 Q-network of /root/reference/BS_brain.py:17-239 (GNNLayer, AggLayer,
huber_loss, BS) with hand-written gfx950 HIP kernels reached through the C ABI in
include/v2xgnn.h.  There is NO CPU fallback: importing works anywhere (so the host logic is
testable), but constructing a model without the HIP library or without a GPU raises.
"""
from .spec import GnnSpec                                     # noqa: F401
from .packing import (pack_xe, adj_to_csr, kron_to_adj, feed_to_arrays, PackedBatch,   # noqa: F401
                      keras_list_to_flat, flat_to_keras_list, keras_list_shapes)
from .lib import load_library, library_path, V2XError         # noqa: F401
from .engine import GnnEngine                                 # noqa: F401
from .bs_brain import BS, GnnQModel, History                  # noqa: F401

__all__ = ["GnnSpec", "BS", "GnnQModel", "History", "GnnEngine", "PackedBatch", "pack_xe", "adj_to_csr",
           "kron_to_adj", "feed_to_arrays", "keras_list_to_flat", "flat_to_keras_list",
           "keras_list_shapes", "load_library", "library_path", "V2XError"]
