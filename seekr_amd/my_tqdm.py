"""Progress-bar selection (reference: seekr/my_tqdm.py:17-32): notebook bar inside a Jupyter
kernel, console bar otherwise; a no-op shim when tqdm is not installed."""
import sys


def _in_notebook():
    if "IPython" not in sys.modules:
        return False
    try:
        from IPython import get_ipython
        return "IPKernelApp" in get_ipython().config
    except Exception:  # noqa: BLE001
        return False


def my_tqdm():
    try:
        if _in_notebook():
            from tqdm import tqdm_notebook as bar
        else:
            from tqdm import tqdm as bar
        return bar
    except ImportError:
        def bar(iterable=None, **_kw):
            return iterable
        return bar


_is_kernel = _in_notebook  # the reference's name for the same question (my_tqdm.py:17-25)


def my_trange():
    """The range-shaped twin (my_tqdm.py:32-33): `tnrange` inside a Jupyter kernel, `trange` otherwise, plain `range`
    when tqdm is not installed."""
    try:
        if _in_notebook():
            from tqdm import tnrange as rng
        else:
            from tqdm import trange as rng
        return rng
    except ImportError:
        def rng(*args, **_kw):
            return range(*args)
        return rng
