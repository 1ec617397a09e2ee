from . import meanshift_cluster  # noqa: F401
