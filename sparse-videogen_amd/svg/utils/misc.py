"""ref: svg/utils/misc.py — terminal colours used by log lines."""


class Color:
    red = "\033[31m"
    green = "\033[32m"
    yellow = "\033[33m"
    blue = "\033[34m"
    reset = "\033[0m"
