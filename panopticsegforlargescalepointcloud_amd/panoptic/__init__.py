from .structures import PanopticLabels, PanopticResults, non_max_suppression  # noqa: F401
from .pointgroup3heads import PointGroup3heads  # noqa: F401
from .variants import PointGroup, PointGroupEmbed  # noqa: F401

# `class:` values of conf/models/panoptic/*.yaml ("<module>.<Class>", resolved by the reference's model factory,
# torch_points3d/models/model_factory.py) -> build-owned classes
MODEL_CLASSES = {
    "PointGroup3heads.PointGroup3heads": PointGroup3heads,
    "pointgroup.PointGroup": PointGroup,
    "pointgroupembed.PointGroupEmbed": PointGroupEmbed,
}


def instantiate_model(cfg, dataset, model_type="dummy", modules=None):
    """The reference's `instantiate_model` for the panoptic task: picks the class named by cfg["class"]."""
    try:
        cls = MODEL_CLASSES[cfg["class"]]
    except KeyError:
        raise NotImplementedError("model class %r (panoptic hot path provides: %s)" % (cfg.get("class"), sorted(MODEL_CLASSES)))
    return cls(cfg, model_type, dataset, modules)
