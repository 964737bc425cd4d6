import torch


class GloVe:
    """No pretrained vectors offline: an empty vocabulary, so every word takes the reference's 'unknown word' branch."""

    def __init__(self, name=None, dim=300, cache=None, max_vectors=None):
        self.stoi = {}
        self.vectors = torch.zeros(1, dim)
