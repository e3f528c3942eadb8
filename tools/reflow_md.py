#!/usr/bin/env python3
"""Reflow a Markdown file to a column limit (default 160): paragraphs and list items are re-wrapped; a table one of whose rows
is longer than the limit becomes a list (first cell bold, the other cells joined with ' — ', header cells as labels).  Code
blocks and short tables stay as they are.  Usage: tools/reflow_md.py FILE [limit]"""
import re
import sys
import textwrap


def wrap(text, indent, first_indent, limit):
    return textwrap.fill(" ".join(text.split()), width=limit, initial_indent=first_indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False)


def main(path, limit=160):
    lines = open(path).read().split("\n")
    out, i = [], 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("```"):
            out.append(l); i += 1
            while i < len(lines) and not lines[i].startswith("```"):
                out.append(lines[i]); i += 1
            if i < len(lines):
                out.append(lines[i]); i += 1
            continue
        if l.startswith("|"):
            blk = []
            while i < len(lines) and lines[i].startswith("|"):
                blk.append(lines[i]); i += 1
            if max(len(x) for x in blk) <= limit:
                out += blk
                continue
            rows = [[c.strip() for c in r.strip().strip("|").split("|")] for r in blk]
            head = rows[0] if len(rows) > 1 and re.match(r"^[\s|:-]+$", blk[1]) else None
            body = rows[2:] if head else rows
            for r in body:
                cells = [c for c in r]
                first = cells[0]
                rest = []
                for k, c in enumerate(cells[1:], 1):
                    if not c:
                        continue
                    label = head[k] if head and k < len(head) and head[k] else ""
                    rest.append(("*%s:* " % label if label else "") + c)
                item = ("**%s**" % first if first else "") + ((" — " if first else "") + " — ".join(rest) if rest else "")
                out.append(wrap(item, "  ", "* ", limit))
            out.append("")
            continue
        if not l.strip() or l.startswith("#") or l.startswith("---"):
            out.append(l); i += 1
            continue
        m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", l)
        if m:
            ind = " " * len(m.group(0))
            para = [l[len(m.group(0)):]]
            i += 1
            while i < len(lines) and lines[i].strip() and not re.match(r"^\s*([*+-]|\d+\.)\s+", lines[i]) and not lines[i].startswith(("#", "|", "```")):
                para.append(lines[i].strip()); i += 1
            out.append(wrap(" ".join(para), ind, m.group(0), limit))
            continue
        para = [l]
        i += 1
        while i < len(lines) and lines[i].strip() and not re.match(r"^\s*([*+-]|\d+\.)\s+", lines[i]) and not lines[i].startswith(("#", "|", "```")):
            para.append(lines[i]); i += 1
        out.append(wrap(" ".join(para), "", "", limit))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 160)
