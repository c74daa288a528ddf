"""On an execute_model exception, log the config, an anonymised view of the
scheduler output and the scheduler stats (reference: vllm/logging_utils/dump_input.py:57-81)."""
from __future__ import annotations

from ..logger import init_logger

logger = init_logger(__name__)


def _anon(so) -> dict:
    return {
        "new_reqs": [{"req_id": r.req_id, "prompt_len": len(r.prompt_token_ids),
                      "num_computed": r.num_computed_tokens, "blocks": [len(b) for b in r.block_ids]}
                     for r in so.scheduled_new_reqs],
        "cached_reqs": list(so.scheduled_cached_reqs.req_ids),
        "num_scheduled_tokens": dict(so.num_scheduled_tokens),
        "total": so.total_num_scheduled_tokens,
        "spec": {k: len(v) for k, v in so.scheduled_spec_decode_tokens.items()},
        "finished": sorted(so.finished_req_ids),
    }


def dump_engine_exception(vllm_config, scheduler_output, scheduler_stats) -> None:
    try:
        logger.error("engine step failed; config: %s", vllm_config)
        logger.error("scheduler output: %s", _anon(scheduler_output))
        if scheduler_stats is not None:
            logger.error("scheduler stats: %s", scheduler_stats)
    except Exception:   # never mask the original error
        logger.exception("failed to dump engine state")
