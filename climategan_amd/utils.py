"""The helpers of the reference's ``climategan/utils.py`` that the hot path needs."""


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


def flatten_opts(opts) -> dict:
    """reference utils.py:385-427: a nested dict -> one level, keys joined by '.'; lists of dicts are indexed (``a.0.b``),
    other lists become their ``str``; how ``Trainer.eval_images`` prints / logs its metric table (trainer.py:1791-1797)."""
    from pathlib import Path

    out = {}

    def walk(d, prefix):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v, prefix + str(k) + ".")
            elif isinstance(v, list):
                if v and isinstance(v[0], dict):
                    for i, m in enumerate(v):
                        walk(m, prefix + str(k) + "." + str(i) + ".")
                else:
                    out[prefix + str(k)] = str(v)
            else:
                out[prefix + str(k)] = str(v) if isinstance(v, Path) else v

    walk(opts, "")
    return out
