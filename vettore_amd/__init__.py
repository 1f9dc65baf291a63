"""vettore_amd -- MI355X-native flat vector search behind Vettore's flat-index
surface.  `nifs` mirrors `Vettore.Nifs`, `index_flat` mirrors
`Vettore.Index.Flat` / the `Vettore.Index` behaviour, `collection` the caller
side of the hot path; all of them drive libvettore_hip.so (HIP, gfx950)."""
from . import _lib

_lib.load()  # fails loudly when the HIP library is missing: there is no fallback

__all__ = ["_lib", "nifs", "index_flat", "collection", "sharded"]
