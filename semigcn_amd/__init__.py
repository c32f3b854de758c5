"""semigcn_amd -- MI355X (gfx950) implementation of SeMIGCN's graph-convolution
message-passing hot path behind the reference's torch.nn.Module surface.

    from semigcn_amd.nn import ChebConv, GCNConv, Sequential        # operator tier
    from semigcn_amd.networks import SingleScaleGCN                  # model tier (SGCN)
    from semigcn_amd.meshnet import MGCN, MeshPool, MeshUnpool       # model tier (MGCN)

Every aggregation / pooling call goes through libsemigcn_hip.so (include/semigcn.h);
there is no CPU or PyTorch fallback for them.
"""
from . import capi  # noqa: F401

__all__ = ["capi", "nn", "networks", "meshnet", "graph", "functional", "synth", "compat"]
__version__ = "0.1.0"
