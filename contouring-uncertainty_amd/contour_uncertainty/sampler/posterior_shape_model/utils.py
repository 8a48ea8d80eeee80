"""reference sampler/posterior_shape_model/utils.py:4-25"""


def index_to_flat(indices):
    """[0] for (K, 2) -> [0, 1] for (2K,)"""
    if isinstance(indices, int):
        return [indices * 2, indices * 2 + 1]
    out = []
    for idx in indices:
        out.extend([idx * 2, idx * 2 + 1])
    return out
