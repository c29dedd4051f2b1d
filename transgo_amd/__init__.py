"""transgo_amd -- MI355X-native batched MCTS self-play engine for Go (drop-in for Transgo's self-play hot path)."""
from ._lib import TransgoError, load  # noqa: F401
