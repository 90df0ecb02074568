"""Import stub (build container only): the reference's loader imports mmcv at module scope."""


class Config(dict):
    __getattr__ = dict.get
