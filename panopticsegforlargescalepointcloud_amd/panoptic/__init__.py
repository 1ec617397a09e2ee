from .structures import PanopticLabels, PanopticResults, non_max_suppression  # noqa: F401
from .pointgroup3heads import PointGroup3heads  # noqa: F401
from .variants import PointGroup, PointGroupEmbed  # noqa: F401
