"""META_ARCHITECTURE + build_model, as /root/reference/builders/model_builder.py:1-10."""
import torch

from .registry import Registry

META_ARCHITECTURE = Registry("ARCHITECTURE")


def build_model(config, vocab):
    model = META_ARCHITECTURE.get(config.ARCHITECTURE)(config, vocab)
    return model.to(torch.device(config.DEVICE))
