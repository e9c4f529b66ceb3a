#!/usr/bin/env python3
"""Re-wrap a markdown file to <= WIDTH columns: paragraphs and list items are hard-wrapped, tables whose rows exceed WIDTH become
definition-style lists (first cell bold, the other cells as wrapped continuation paragraphs labelled with their column header); code
blocks and short tables are kept.   python dev/rewrap_md.py IN.md OUT.md [WIDTH]"""
import re
import sys
import textwrap


def wrap(text, width, first="", rest=""):
    return textwrap.fill(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    return [c.strip() for c in re.split(r"(?<!\\)\|", row)]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 190
    lines = open(src).read().split("\n")
    out, i, in_code = [], 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("```"):
            in_code = not in_code
            out.append(ln)
            i += 1
            continue
        if in_code:
            out.append(ln)
            i += 1
            continue
        if ln.startswith("|") and i + 1 < len(lines) and re.match(r"^\|[\s:|-]+\|?$", lines[i + 1].strip()):
            j = i
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            tbl = lines[i:j]
            if max(len(t) for t in tbl) <= width:
                out.extend(tbl)
            else:
                head = cells(tbl[0])
                for row in tbl[2:]:
                    c = cells(row)
                    out.append(wrap(f"**{c[0]}**" if c and c[0] else "**-**", width, "- ", "  "))
                    for h, v in zip(head[1:], c[1:]):
                        if v:
                            out.append(wrap(f"*{h}:* {v}" if h else v, width, "  - ", "    "))
                out.append("")
            i = j
            continue
        m = re.match(r"^(\s*)([-*]|\d+\.)\s+(.*)$", ln)
        if m and len(ln) > width:
            ind = m.group(1) + m.group(2) + " "
            out.append(wrap(m.group(3), width, ind, " " * len(ind)))
        elif len(ln) > width and not ln.startswith("#"):
            out.append(wrap(ln, width))
        else:
            out.append(ln)
        i += 1
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
