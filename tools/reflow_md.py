"""Paragraph-level re-flow of a markdown file to WIDTH columns: the lines of a paragraph or list item are joined and filled again (tools/wrap_md.py only cuts lines that are
too long, which leaves ragged paragraphs after edits).  Tables, code fences, headings and lines that end in two spaces are left alone.
    python tools/reflow_md.py file.md [width]"""
import re
import sys
import textwrap

WIDTH = int(sys.argv[2]) if len(sys.argv) > 2 else 160
ITEM = re.compile(r"^(\s*)([*-]|\d+\.)\s+")


def fill(first_prefix, text, indent):
    return textwrap.fill(text, width=WIDTH, initial_indent=first_prefix, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False)


def flush(buf, out):
    if not buf:
        return
    m = ITEM.match(buf[0])
    if m:
        prefix = buf[0][:m.end()]
        indent = " " * len(prefix)
        text = " ".join([buf[0][m.end():].strip()] + [b.strip() for b in buf[1:]])
    else:
        lead = re.match(r"^\s*", buf[0]).group(0)
        prefix = indent = lead
        text = " ".join(b.strip() for b in buf)
    out.append(fill(prefix, text, indent))
    buf.clear()


def main():
    lines = open(sys.argv[1], encoding="utf-8").read().split("\n")
    out, buf, fence = [], [], False
    for l in lines:
        if l.strip().startswith("```"):
            flush(buf, out)
            fence = not fence
            out.append(l)
            continue
        if fence or l.startswith("#") or l.lstrip().startswith("|") or not l.strip():
            flush(buf, out)
            out.append(l)
            continue
        if ITEM.match(l):
            flush(buf, out)
        buf.append(l)
    flush(buf, out)
    open(sys.argv[1], "w", encoding="utf-8").write("\n".join(out))


if __name__ == "__main__":
    main()
