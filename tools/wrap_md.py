"""Re-flow a markdown file to a column limit: paragraphs and list items are joined and wrapped (continuation lines indented under the
item), fenced code, headings and short tables are left alone; with no --keep-tables a table whose rows exceed the limit is turned into one
small section per row (first cell = heading, the other cells = bullets labelled with their column names) -- a 3 000-character table cell
cannot be read in any viewer.   python tools/wrap_md.py IN OUT [120] [--keep-tables]"""
import re
import sys
import textwrap

args = [a for a in sys.argv[1:] if not a.startswith("--")]
src, dst = args[0], args[1]
width = int(args[2]) if len(args) > 2 else 120
keep_tables = "--keep-tables" in sys.argv
lines = open(src).read().split("\n")
out, i = [], 0


def wrap(text, first, rest):
    return textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


LIST = re.compile(r"^(\s*(?:[-*]|\d+\.)\s+)(.*)$")


def special(l):
    return (not l.strip()) or l.lstrip().startswith("```") or l.startswith("|") or l.startswith("#") or LIST.match(l) or l.startswith("<!--")


while i < len(lines):
    l = lines[i]
    if l.lstrip().startswith("```"):                       # fenced code: verbatim
        out.append(l); i += 1
        while i < len(lines) and not lines[i].lstrip().startswith("```"):
            out.append(lines[i]); i += 1
        if i < len(lines):
            out.append(lines[i]); i += 1
        continue
    if not l.strip() or l.startswith("#") or l.startswith("<!--"):
        out.append(l); i += 1; continue
    if l.startswith("|"):
        j = i
        while j < len(lines) and lines[j].startswith("|"):
            j += 1
        block = lines[i:j]
        if keep_tables or all(len(b) <= width for b in block):
            out += block
        else:
            split = lambda row: [c.strip() for c in re.split(r"\s\|\s", row.strip().strip("|").strip())]
            head = split(block[0])
            for row in block[2:]:
                c = split(row)
                out.append("")
                out += wrap(c[0], "**", "  ")
                out[-1] += "**"
                for name, cell in zip(head[1:], c[1:]):
                    if cell:
                        out += wrap(f"{name}: {cell}", "  - ", "    ")
            out.append("")
        i = j
        continue
    m = LIST.match(l)
    if m:
        first, text = m.group(1), m.group(2)
        i += 1
        while i < len(lines) and not special(lines[i]) and lines[i].startswith(" "):      # continuation lines of the item
            text += " " + lines[i].strip(); i += 1
        out += wrap(text, first, " " * len(first))
        continue
    ind = re.match(r"^\s*", l).group(0)
    text = l.strip()
    i += 1
    while i < len(lines) and not special(lines[i]):
        text += " " + lines[i].strip(); i += 1
    out += wrap(text, ind, ind)
open(dst, "w").write("\n".join(out))
print(max(len(x) for x in out), "max columns;", len(out), "lines")
