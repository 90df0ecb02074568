"""Import stub (build container only)."""


class WarmupCosineLR:
    pass


class WarmupMultiStepLR:
    pass
