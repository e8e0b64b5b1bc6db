"""placeholder (replaced below in this round)"""
from torch import nn


class CSNet(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError
