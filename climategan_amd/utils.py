"""The two helpers of the reference's ``climategan/utils.py`` that the hot path needs."""


def find_target_size(opts, task):
    """reference utils.py:984-995: final ``resize`` transform's new_size for ``task`` (int or per-task dict)."""
    try:
        new_size = opts.data.transforms[-1].new_size
    except (AttributeError, KeyError, IndexError, TypeError):
        return None
    if isinstance(new_size, int):
        return new_size
    if not new_size:
        return None
    if task in new_size:
        return new_size[task]
    assert "default" in new_size
    return new_size["default"]
