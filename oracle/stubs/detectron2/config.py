"""Import stub (build container only)."""


class CfgNode(dict):
    pass
