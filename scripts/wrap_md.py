"""Re-wrap the prose lines of a markdown file to a width (tables, headings and fenced code are left alone).
usage: python scripts/wrap_md.py FILE [WIDTH=118]"""
import re
import sys
import textwrap


def wrap(text, width):
    out, fence = [], False
    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
        if fence or len(line) <= width or line.lstrip().startswith(("|", "#")):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[*\-+]|\d+\.)\s+)?", line)
        lead, bullet = m.group(1), m.group(2) or ""
        body = line[len(lead) + len(bullet):]
        out.extend(textwrap.wrap(body, width=width, initial_indent=lead + bullet, subsequent_indent=lead + " " * len(bullet),
                                 break_long_words=False, break_on_hyphens=False))
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    src = open(path).read()
    open(path, "w").write(wrap(src, width))
