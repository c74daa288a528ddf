from .layerwise import LayerwiseProfiler, LayerwiseProfileResults, layerwise_profile  # noqa: F401
