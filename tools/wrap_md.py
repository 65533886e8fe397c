#!/usr/bin/env python
"""Re-wrap the prose of a markdown file at WIDTH columns (tables, code fences and headings are left alone; list items keep their hanging
indent).   python tools/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap


def wrap(text, width=150):
    out, fence = [], False
    for line in text.split('\n'):
        if line.lstrip().startswith('```'):
            fence = not fence
            out.append(line)
            continue
        if fence or len(line) <= width or line.lstrip().startswith('|') or line.startswith('#'):
            out.append(line)
            continue
        m = re.match(r'^(\s*(?:[-*+]|\d+\.)\s+)', line)
        lead = re.match(r'^\s*', line).group(0)
        first, rest = (m.group(1), ' ' * len(m.group(1))) if m else (lead, lead)
        body = line[len(first):]
        out.extend(textwrap.wrap(body, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False))
    return '\n'.join(out)


if __name__ == '__main__':
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    s = open(path).read()
    open(path, 'w').write(wrap(s, width))
