"""Side channels between the fused layers (pure Python; no device code)."""


# ----------------------------------------------------------- side channels between fused layers
# A fused layer hands its consumer the channel-last twin of its (B, C, N) output (and the
# sampling prefetch hands the layers what it derived from the coordinates) as an attribute of
# the tensor.  Every such attribute carries the `_version` of the tensor(s) it was derived from
# and is ignored once one of them was written to in place: a stale twin would be a silent
# wrong result.
def attach_twin(t, twin):
    t._btr_channel_last = (twin, t._version)


def twin_of(t):
    """The channel-last twin attached to `t`, or None (never attached, or `t` modified since)."""
    pair = getattr(t, "_btr_channel_last", None) if t is not None else None
    if pair is None or pair[1] != t._version:
        return None
    return pair[0]


def attach_derived(t, name, value, *sources):
    """t.<name> = value derived from `t` and `sources` (their versions are remembered)."""
    setattr(t, name, (value, tuple(s._version for s in (t,) + sources)))


def derived(t, name, *sources):
    """The value attached by attach_derived if `t` and `sources` are unmodified since."""
    pair = getattr(t, name, None)
    if pair is None or pair[1] != tuple(s._version for s in (t,) + sources):
        return None
    return pair[0]
