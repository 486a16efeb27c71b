"""Node API and pull protocol -- the drop-in boundary.

Same names, ports, state attributes, error types and `(frames, channels)` semantics as the
reference's `signals.chain` (reference src/signals/chain/__init__.py:21-457), written fresh with
`torch.Tensor` buffers resident on the MI355X in place of numpy arrays:

  * a reply with one row (block rate, `forward_at_block_rate`) is a float64 control row;
  * a reply with more rows is float32 audio (arithmetic inside the kernels is float64 where the
    1e-6 bar needs it -- phase, filter coefficients and state);
  * replies are borrowed, read-only views: cache hits return the cached tensor or a slice of it,
    `Fixed` returns its resident upload.

Layout: `blocks` (Shape / BlockLoc / Request / errors), `nodes` (Signal / Emitter / Receiver / port /
mixins), `cache` (BlockCachingEmitter).  Nothing in this package computes samples; the `_eval`
bodies in osc / fx / noise / shape / ext call the HIP kernels through `signals_amd._native`.
"""
from signals_amd.chain.blocks import (  # noqa: F401
    BadShape,
    BadStateSchema,
    BadStateValue,
    BlockLoc,
    ChainLayerError,
    Request,
    RequestRate,
    Shape,
)
from signals_amd.chain.nodes import (  # noqa: F401
    AUDIO_DTYPE,
    CTRL_DTYPE,
    BoundPort,
    Emitter,
    ExplicitChannels,
    ExplicitChannelsEmitter,
    HostSnapshot,
    ImplicitChannels,
    PassThroughResult,
    Receiver,
    Signal,
    adopt_reply,
    as_control,
    broadcast_shape,
    concatenate,
    graph_clock,
    host_plugins,
    port,
    result_dtype,
    state,
)
from signals_amd.chain.cache import BlockCachingEmitter, NotCached  # noqa: F401
