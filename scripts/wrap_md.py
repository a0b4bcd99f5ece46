"""Re-flow the prose of a markdown file to a width: paragraphs and list items are joined and wrapped again; tables, headings,
fenced and indented code and blank lines are left alone.  usage: python scripts/wrap_md.py FILE [WIDTH=118]"""
import re
import sys
import textwrap

BULLET = re.compile(r"^(\s*)((?:[*\-+]|\d+\.)\s+)")


def reflow(text, width):
    lines = text.split("\n")
    out, i, fence = [], 0, False
    prev_blank = True
    while i < len(lines):
        line = lines[i]
        if line.lstrip().startswith("```"):
            fence = not fence
            out.append(line)
            i += 1
            prev_blank = False
            continue
        if fence or not line.strip() or line.lstrip().startswith(("|", "#")) or (prev_blank and line.startswith("    ") and not BULLET.match(line)):
            out.append(line)
            prev_blank = not line.strip()
            i += 1
            continue
        m = BULLET.match(line)
        lead = m.group(1) if m else re.match(r"^\s*", line).group(0)
        bullet = m.group(2) if m else ""
        body = [line[len(lead) + len(bullet):].strip()]
        cont = lead + " " * len(bullet)
        i += 1
        while i < len(lines):
            nxt = lines[i]
            if not nxt.strip() or nxt.lstrip().startswith(("|", "#", "```")) or BULLET.match(nxt):
                break
            if bullet and not nxt.startswith(cont[:1] if cont else "") and cont:
                break
            body.append(nxt.strip())
            i += 1
        out.extend(textwrap.wrap(" ".join(body), width=width, initial_indent=lead + bullet, subsequent_indent=cont,
                                 break_long_words=False, break_on_hyphens=False))
        prev_blank = False
    return "\n".join(out)


if __name__ == "__main__":
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
    src = open(path).read()
    open(path, "w").write(reflow(src, width))
