"""MI355X-native LLM serving engine.

Public names mirror the reference so user scripts keep working:
``LLM``, ``SamplingParams``, ``RequestOutput``, ``AsyncLLMEngine`` ...
Heavy submodules are imported lazily so ``import vllm_distributed_amd`` stays cheap.
"""
from .version import __version__

_LAZY = {
    "LLM": ".entrypoints.llm",
    "SamplingParams": ".sampling_params",
    "PoolingParams": ".sampling_params",
    "BeamSearchParams": ".sampling_params",
    "GuidedDecodingParams": ".sampling_params",
    "RequestOutputKind": ".sampling_params",
    "RequestOutput": ".outputs",
    "CompletionOutput": ".outputs",
    "PoolingRequestOutput": ".outputs",
    "EmbeddingRequestOutput": ".outputs",
    "ClassificationRequestOutput": ".outputs",
    "ScoringRequestOutput": ".outputs",
    "EngineArgs": ".engine.arg_utils",
    "AsyncEngineArgs": ".engine.arg_utils",
    "LLMEngine": ".engine.llm_engine",
    "AsyncLLM": ".engine.async_llm",
    "AsyncLLMEngine": ".engine.async_llm",
    "LoRARequest": ".request",
    "TokensPrompt": ".inputs",
    "TextPrompt": ".inputs",
    "ModelRegistry": ".models.registry",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(_LAZY[name], __name__), name)
    raise AttributeError(name)


__all__ = ["__version__", *_LAZY]
