"""Import stub (build container only): the reference's logger imports termcolor at module scope."""


def colored(text, *args, **kwargs):
    return text
