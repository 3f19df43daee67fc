"""Terminal colours and the print helpers the reference's entry scripts import from `svg.utils.misc` (ref: svg/utils/misc.py;
wan_i2v_inference.py:18).  Same names, same escape sequences."""
import torch

_SGR_NAMES = ("black", "red", "green", "yellow", "blue", "magenta", "cyan", "white")   # ANSI SGR 30 .. 37, in that order


def _sgr(code: str) -> str:
    return f"\033[{code}m"


# `Color.red`, `Color.reset`, ...: the eight standard foreground colours, the default foreground (39) and one 24-bit orange
Color = type("Color", (), {**{name: _sgr(str(30 + i)) for i, name in enumerate(_SGR_NAMES)},
                           "reset": _sgr("39"), "orange": _sgr("38;2;180;60;0")})


def _hbm_mb():
    """(allocated, peak) of the current device in MB; zeros without a GPU"""
    if not torch.cuda.is_available():
        return 0, 0
    return tuple(n // 1024 ** 2 for n in (torch.cuda.memory_allocated(), torch.cuda.max_memory_allocated()))


def clear_memory_usage():
    """drop the caching allocator's free blocks and restart the peak counter (ref :20-22)"""
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


def print_memory_usage(prefix: str = ""):
    """`<prefix> Memory: <allocated> / <peak> MB` in orange (ref :25-28)"""
    cur, peak = _hbm_mb()
    print(Color.orange + f"{prefix} Memory: {cur} / {peak} MB" + Color.reset)


def print_args(args):
    """the parsed command line, one `name: value` per line, in magenta (ref :31-34)"""
    lines = ["Args:"] + [f"{k}: {v}" for k, v in vars(args).items()]
    print("\n".join(Color.magenta + line + Color.reset for line in lines))
