"""Re-flow a markdown file to a column limit: paragraphs and list items are wrapped (continuation lines indented under the item), fenced code
is left alone, and a table whose rows exceed the limit is turned into one small section per row (first cell = heading, the other cells =
bullets labelled with their column names) -- a 3 000-character table cell cannot be read in any viewer.   python tools/wrap_md.py IN OUT [120]"""
import re
import sys
import textwrap

src, dst = sys.argv[1], sys.argv[2]
width = int(sys.argv[3]) if len(sys.argv) > 3 else 120
lines = open(src).read().split("\n")
out, i, in_code = [], 0, False


def wrap(text, first, rest):
    return textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False) or [first.rstrip()]


def cells(row):
    return [c.strip() for c in row.strip().strip("|").split("|")]


while i < len(lines):
    l = lines[i]
    if l.lstrip().startswith("```"):
        in_code = not in_code
        out.append(l); i += 1; continue
    if in_code or len(l) <= width and not l.startswith("|"):
        out.append(l); i += 1; continue
    if l.startswith("|"):
        j = i
        while j < len(lines) and lines[j].startswith("|"):
            j += 1
        block = lines[i:j]
        if all(len(b) <= width for b in block):
            out += block
        else:
            # cells may contain escaped or code-quoted pipes: split on " | " only
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
    m = re.match(r"^(\s*(?:[-*]|\d+\.)\s+)(.*)$", l)
    if m:
        out += wrap(m.group(2), m.group(1), " " * len(m.group(1)))
    else:
        ind = re.match(r"^\s*", l).group(0)
        out += wrap(l.strip(), ind, ind)
    i += 1
open(dst, "w").write("\n".join(out))
print(max(len(x) for x in out), "max columns;", len(out), "lines")
