"""Import stub (build container only): the reference's loader imports mmcv at module scope, and tools/training_utils.py:33,55
wraps its solver settings in mmcv.Config (a recursive attribute dictionary: cfg.SOLVER.OPTIMIZER_CFG, cfg.SOLVER.get(...))."""


class Config(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, Config):
                self[key] = Config(v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            return None

    def __setattr__(self, name, value):
        self[name] = value
