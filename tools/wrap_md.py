"""Re-wrap the prose of a markdown file at ~120 columns (tables, code fences and headings are left alone; list items keep a
hanging indent).   python tools/wrap_md.py DESIGN.md [width]"""
import re
import sys
import textwrap


def wrap(text, width=120):
    out, fence = [], False
    for line in text.split('\n'):
        if line.lstrip().startswith('```'):
            fence = not fence
            out.append(line)
            continue
        if fence or len(line) <= width or line.lstrip().startswith(('|', '#')):
            out.append(line)
            continue
        m = re.match(r'^(\s*(?:[-*]|\d+\.)\s+|\s*)', line)
        lead = m.group(1)
        hang = ' ' * len(lead)
        out.extend(textwrap.wrap(line, width=width, initial_indent='', subsequent_indent=hang, break_long_words=False,
                                 break_on_hyphens=False))
    return '\n'.join(out)


if __name__ == '__main__':
    p = sys.argv[1]
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    s = open(p).read()
    open(p, 'w').write(wrap(s, w))
