"""MI355X-native self-attention style discriminator (hot path of
0e4e6d01/non-parallel-text-style-transfer-using-self-attn-discriminator).

Public surface (lazy: importing the package does not import torch or load the HIP library):

    sad_amd.models.SelfAttnRNNClassifier   drop-in for the reference class of the same name
    sad_amd.functional                     autograd bridge to libsad.so (C-ABI, include/sad.h)
    sad_amd.lib                            ctypes loader / error mapping
    sad_amd.dp                             batch data-parallel helpers (torch.distributed / RCCL)
    sad_amd.synth                          seeded numpy generators for weights and batches
"""
__version__ = "0.1.0"
__all__ = ["models", "functional", "lib", "dp", "synth", "SelfAttnRNNClassifier"]


def __getattr__(name):
    import importlib
    if name == "SelfAttnRNNClassifier":
        return importlib.import_module(__name__ + ".models").SelfAttnRNNClassifier
    if name in __all__:
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
