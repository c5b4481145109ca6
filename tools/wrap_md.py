"""Re-flow a markdown file to lines of at most WIDTH columns (DESIGN.md had lines of 1 400+ characters).

    python tools/wrap_md.py in.md out.md [width]

Paragraphs and list items are wrapped (continuation lines indented under the item's text); code fences and short tables are left alone; a table with a row wider
than WIDTH becomes a nested list — one item per row, headed by the row's first cell, one sub-item per remaining non-empty cell, labelled with its column header —
because a table row cannot be wrapped.  Headings longer than WIDTH are left as they are (they must stay on one line)."""
import re
import sys
import textwrap

WIDTH = 160


def wrap_item(prefix, text, width):
    ind = " " * len(prefix)
    return textwrap.fill(text, width=width, initial_indent=prefix, subsequent_indent=ind, break_long_words=False, break_on_hyphens=False)


def cells(row):
    row = row.strip()
    if row.startswith("|"):
        row = row[1:]
    if row.endswith("|"):
        row = row[:-1]
    out, cur, code = [], "", False          # a pipe inside `code` or escaped as \| does not split
    i = 0
    while i < len(row):
        ch = row[i]
        if ch == "`":
            code = not code
        if ch == "\\" and i + 1 < len(row) and row[i + 1] == "|":
            cur += "|"; i += 2; continue
        if ch == "|" and not code:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
        i += 1
    out.append(cur.strip())
    return out


def convert(lines, width=WIDTH):
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i].rstrip("\n")
        if ln.lstrip().startswith("```"):                       # code fence: verbatim
            out.append(ln); i += 1
            while i < n and not lines[i].lstrip().startswith("```"):
                out.append(lines[i].rstrip("\n")); i += 1
            if i < n:
                out.append(lines[i].rstrip("\n")); i += 1
            continue
        if ln.lstrip().startswith("|") and i + 1 < n and re.match(r"^\s*\|?\s*:?-{2,}", lines[i + 1]):      # table
            j = i
            while j < n and lines[j].lstrip().startswith("|"):
                j += 1
            block = [l.rstrip("\n") for l in lines[i:j]]
            if max(len(b) for b in block) <= width:
                out.extend(block)
            else:
                head = cells(block[0])
                for row in block[2:]:
                    c = cells(row)
                    first = c[0] if c and c[0] else "(row)"
                    out.append(wrap_item("- ", "**%s** %s" % (head[0], first) if head and head[0] else first, width))
                    for k in range(1, len(c)):
                        if c[k]:
                            label = head[k] if k < len(head) and head[k] else "col %d" % k
                            out.append(wrap_item("  - ", "*%s*: %s" % (label, c[k]), width))
                out.append("")
            i = j
            continue
        if not ln.strip() or ln.startswith("#") or len(ln) <= width:
            out.append(ln); i += 1
            continue
        m = re.match(r"^(\s*(?:[-*+]|\d+\.)\s+|\s*>\s?|\s+)?(.*)$", ln)
        prefix, text = m.group(1) or "", m.group(2)
        out.append(wrap_item(prefix, text, width))
        i += 1
    return out


if __name__ == "__main__":
    w = int(sys.argv[3]) if len(sys.argv) > 3 else WIDTH
    res = convert(open(sys.argv[1]).read().split("\n"), w)
    open(sys.argv[2], "w").write("\n".join(res))
    over = [k + 1 for k, l in enumerate(res) if len(l) > w and not l.startswith("#")]
    print("%s: %d lines, %d over %d columns %s" % (sys.argv[2], len(res), len(over), w, over[:10]))
