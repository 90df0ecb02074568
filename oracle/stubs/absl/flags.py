"""Minimal stand-in for absl.flags: an attribute bag plus no-op DEFINE_* helpers."""


class _Flags:
    def __init__(self):
        object.__setattr__(self, "_d", {})

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_d")[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        object.__getattribute__(self, "_d")[k] = v


FLAGS = _Flags()


def _define(name, default, help="", **kw):
    if name not in object.__getattribute__(FLAGS, "_d"):
        setattr(FLAGS, name, default)


DEFINE_integer = DEFINE_float = DEFINE_string = DEFINE_bool = DEFINE_boolean = _define
