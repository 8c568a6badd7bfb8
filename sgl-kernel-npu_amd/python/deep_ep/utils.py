"""EventOverlap + the debug logging decorator (reference python/deep_ep/deep_ep/utils.py:12-113)."""
import functools
import inspect
import logging
from typing import Optional, Tuple

import torch

logger = logging.getLogger("deep_ep")


class EventOverlap:
    """Handle returned next to every communication result.

    Every MI355X op is enqueued on the caller's current stream, so (exactly like the reference, utils.py:32-33)
    there is nothing to wait for; when a native EventHandle is attached, `current_stream_wait()` forwards to it so
    a caller that switched streams is still ordered correctly."""

    def __init__(self, event=None, extra_tensors: Optional[Tuple[torch.Tensor]] = None) -> None:
        self.event = event
        self.extra_tensors = extra_tensors     # keeps tensors alive across async use (CUDA-graph friendly)

    def current_stream_wait(self) -> None:
        if self.event is not None and hasattr(self.event, "current_stream_wait"):
            self.event.current_stream_wait()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.current_stream_wait()


def _brief(v):
    if isinstance(v, (tuple, list)):
        return ", ".join(_brief(a) for a in v)
    if isinstance(v, torch.Tensor):
        return str((v.dtype, tuple(v.shape)))
    return str(v)


def log_parameters(input_name_full_tensor=None, output_idx_full_tensor=None):
    """DEBUG-level call tracing: tensors are summarised as (dtype, shape) unless named in
    `input_name_full_tensor` / indexed in `output_idx_full_tensor`."""
    full_in = set(input_name_full_tensor or [])
    full_out = set(output_idx_full_tensor or [])

    def deco(fn):
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            debug = logger.isEnabledFor(logging.DEBUG)
            if debug:
                bound = sig.bind(*args, **kwargs)
                bound.apply_defaults()
                who = getattr(bound.arguments.get("self"), "rank", "unknown")
                lines = [f"{k}: {v if k in full_in else _brief(v)}" for k, v in bound.arguments.items() if k not in ("self", "cls")]
                logger.debug("[rank %s] calling %s with\n%s", who, fn.__name__, "\n".join(lines))
            out = fn(*args, **kwargs)
            if debug:
                items = out if isinstance(out, tuple) else (out,)
                lines = [str(v) if i in full_out else _brief(v) for i, v in enumerate(items)]
                logger.debug("[rank %s] %s returned\n%s", who, fn.__name__, "\n".join(lines))
            return out

        return wrapper

    return deco
