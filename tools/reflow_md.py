#!/usr/bin/env python3
"""Re-flow the prose paragraphs of a markdown file to a column limit (tables, headings, code fences and list structure are kept).
    python tools/reflow_md.py FILE [width=140]"""
import re
import sys
import textwrap

path, width = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 140
lines = open(path).read().split("\n")
out, para, in_code = [], [], False
item_re = re.compile(r"^(\s*)([-*]|\d+\.)\s+")


def flush():
    global para
    if not para:
        return
    first = para[0]
    m = item_re.match(first)
    if m:
        indent = " " * len(m.group(0))
        text = " ".join(l.strip() for l in para)
        text = m.group(1) + text[len(m.group(1)):] if False else text
        lead = first[: len(m.group(0))]
        body = " ".join([first[len(m.group(0)):].strip()] + [l.strip() for l in para[1:]])
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
    else:
        indent = re.match(r"^\s*", first).group(0)
        body = " ".join(l.strip() for l in para)
        out.extend(textwrap.wrap(body, width=width, initial_indent=indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
    para = []


for l in lines:
    if l.strip().startswith("```"):
        flush()
        in_code = not in_code
        out.append(l)
        continue
    if in_code or not l.strip() or l.lstrip().startswith("|") or l.startswith("#"):
        flush()
        out.append(l)
        continue
    if item_re.match(l):          # a new list item starts a new paragraph
        flush()
        para = [l]
        continue
    if para and item_re.match(para[0]) and not l.startswith(" "):   # an unindented line after a list item: new paragraph
        flush()
    para.append(l)
flush()
open(path, "w").write("\n".join(out))
