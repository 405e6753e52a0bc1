"""Re-wraps the paragraphs and bullets of a Markdown file to at most WIDTH display columns (headings, tables and fenced code are left alone).
usage: tools_wrap_md.py FILE [WIDTH=120]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out, para, indent, first = [], [], "", ""


def flush():
    global para
    if para:
        text = " ".join(x.strip() for x in para)
        text = re.sub(r"(?<=[.;:]) (?=[A-Z(`*\[])", "  ", text) if False else text
        out.extend(textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
        para = []


fence = False
for ln in lines:
    if ln.startswith("```"):
        flush(); fence = not fence; out.append(ln); continue
    if fence or ln.startswith("#") or ln.startswith("|") or not ln.strip():
        flush(); out.append(ln); continue
    m = re.match(r"^(\s*)([*-]|\d+\.)\s+", ln)
    if m:
        flush()
        first = m.group(0)
        indent = " " * len(first)
        para = [ln[len(first):]]
    elif para and (ln.startswith(indent) or not indent):
        para.append(ln)
    else:
        flush()
        first = indent = ""
        para = [ln]
    # a paragraph that follows a bullet without indentation starts afresh
    if not m and para and len(para) == 1 and not ln.startswith(" "):
        first = indent = ""
flush()
open(path, "w").write("\n".join(out))
