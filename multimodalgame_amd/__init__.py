"""multimodalgame_amd -- MI355X-native (gfx950) drop-in for the REINFORCE exchange path of
nyu-dl/MultimodalGame's model.py.  See DESIGN.md / INTEGRATION.md."""
from ._lib import MmgError, LIB_PATH  # noqa: F401
