"""Model plugins (registry group "models")."""
from rltime_amd.general.lazy_types import LazyTypes

_TABLE = LazyTypes({"sequential": "rltime_amd.models.torch.sequential:SequentialModel"})


def get_types():
    return _TABLE
