"""ref: svg/utils/misc.py — terminal colours and the two print helpers the Wan entry script imports
(wan_i2v_inference.py:18)."""
import torch


class Color:
    black = "\033[30m"
    red = "\033[31m"
    green = "\033[32m"
    yellow = "\033[33m"
    blue = "\033[34m"
    magenta = "\033[35m"
    cyan = "\033[36m"
    white = "\033[37m"
    reset = "\033[39m"
    orange = "\033[38;2;180;60;0m"


def clear_memory_usage():
    """ref: svg/utils/misc.py:20-22"""
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


def print_memory_usage(prefix: str = ""):
    """ref: svg/utils/misc.py:25-28 — allocated / peak HBM of the current device in MB"""
    cur, peak = (torch.cuda.memory_allocated(), torch.cuda.max_memory_allocated()) if torch.cuda.is_available() else (0, 0)
    print(f"{Color.orange}{prefix} Memory: {cur // 1024 ** 2} / {peak // 1024 ** 2} MB{Color.reset}")


def print_args(args):
    """ref: svg/utils/misc.py:31-34"""
    print(f"{Color.magenta}Args:{Color.reset}")
    for key, value in args.__dict__.items():
        print(f"{Color.magenta}{key}: {value}{Color.reset}")
