"""finmlkit_amd -- MI355X (gfx950) tick->bar engine behind finmlkit's bar/feature API.

Only the hot path of quantscious/finmlkit is implemented here: bar close
indexers, per-bar OHLCV / order-flow / footprint reducers and the tick-level volatility loops,
all as hand-written HIP kernels in csrc/ reached through the C ABI of include/fmk.h.
"""
__version__ = "0.1.0"
