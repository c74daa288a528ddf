"""Plugin loading through Python entry points.

Group ``vllm.general_plugins``: callables run once per process (engine front end, engine core and
every worker) before the model is built; they typically call ``ModelRegistry.register_model`` or
``KVConnectorFactory.register_connector``.  Group ``vllm.platform_plugins`` is resolved by the
platform layer.  ``VLLM_PLUGINS`` (comma separated names) restricts which plugins run.
Reference: vllm/plugins/__init__.py:16-84.
"""
from __future__ import annotations

from typing import Callable

from .. import envs
from ..logger import init_logger

logger = init_logger(__name__)
DEFAULT_PLUGINS_GROUP = "vllm.general_plugins"
_loaded = False


def load_plugins_by_group(group: str) -> dict[str, Callable]:
    from importlib.metadata import entry_points
    allowed = envs.VLLM_PLUGINS
    try:
        found = entry_points(group=group)
    except TypeError:                       # older importlib.metadata API
        found = entry_points().get(group, [])
    out: dict[str, Callable] = {}
    for ep in found:
        if allowed is not None and ep.name not in allowed:
            continue
        try:
            out[ep.name] = ep.load()
            logger.info("plugin %s loaded from group %s", ep.name, group)
        except Exception:
            logger.exception("failed to load plugin %s", ep.name)
    return out


def load_general_plugins() -> None:
    """Idempotent: safe to call from every process entry point."""
    global _loaded
    if _loaded:
        return
    _loaded = True
    for name, fn in load_plugins_by_group(DEFAULT_PLUGINS_GROUP).items():
        try:
            fn()
        except Exception:
            logger.exception("plugin %s raised while initialising", name)
