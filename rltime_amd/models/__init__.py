"""Model plugins, registry keys as in the reference (rltime/models/__init__.py)."""


def get_types():
    from .torch.sequential import SequentialModel
    return {"sequential": SequentialModel}
