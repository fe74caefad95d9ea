"""tools/rewrap.py FILE... -- bring source lines to at most 140 columns without changing a token (VERDICT round 5, item 3 / Weak 16).

* a pure comment line is cut at the last blank before the limit; the rest continues on a new comment line with the same prefix;
* a code line with a trailing `// comment` gets its comment moved to a line of its own ABOVE the code (same indentation);
* a code line that is still too long is cut at a blank between two tokens -- after `,` / `;` / `{` or before `&&` / `||` / `?` / `:` by
  preference -- and continued with four more columns of indentation (a macro's continuation line gets its ` \\`).
Lines with string literals, preprocessor directives and macro lines that carry a comment are left alone (reported).
C++ is insensitive to white space between tokens, so the token stream -- and with it the code object -- is unchanged; the caller checks
exactly that (build, then compare `llvm-objdump -d` of the code object before and after)."""
import re
import sys

LIMIT = 140


def split_comment(line):
    """(code, comment) at the first // that is not inside a string; comment is None when there is none"""
    in_s = None
    i = 0
    while i < len(line) - 1:
        c = line[i]
        if in_s:
            if c == "\\":
                i += 2
                continue
            if c == in_s:
                in_s = None
        elif c in "\"'":
            in_s = c
        elif c == "/" and line[i + 1] == "/":
            return line[:i], line[i:]
        i += 1
    return line, None


def flow_comment(indent, text, first_prefix="// ", prefix="// "):
    """text (without the leading //) re-flowed into comment lines of at most LIMIT columns"""
    words = text.split(" ")
    out, cur = [], indent + first_prefix.rstrip(" ")
    for w in words:
        if w == "" and cur.endswith(" "):
            cur += " "
            continue
        cand = cur + " " + w
        if len(cand) > LIMIT and cur.strip() not in ("//",):
            out.append(cur.rstrip())
            cur = indent + prefix.rstrip(" ") + " " + w
        else:
            cur = cand
    out.append(cur.rstrip())
    return out


def cut_code(code, cont, last_of_macro=False):
    """code (no comment, no string literal) cut into lines <= LIMIT at blanks between tokens; cont: the line ends with a macro's backslash;
    last_of_macro: it is the LAST line of a macro (the line before it ends with a backslash): every piece but the last needs one"""
    indent = re.match(r"\s*", code).group(0)
    more = indent + "    "
    out = []
    cur = code.rstrip()
    tail = " \\" if (cont or last_of_macro) else ""
    if cont:
        cur = cur[:-1].rstrip()                       # without the backslash
    lim = LIMIT - len(tail)
    def outside_literals(text):
        """positions of text that lie outside string / character literals"""
        ok, in_s, i = [True] * len(text), None, 0
        while i < len(text):
            c = text[i]
            if in_s:
                ok[i] = False
                if c == "\\":
                    if i + 1 < len(text):
                        ok[i + 1] = False
                    i += 2
                    continue
                if c == in_s:
                    in_s = None
            elif c in "\"'":
                in_s = c
                ok[i] = False
            i += 1
        return ok
    while len(cur) > lim:
        seg = cur[:lim + 1]
        free = outside_literals(cur)
        best = -1
        for pat in (r"[,;{] ", r" (?=&&|\|\||\? |: )", r" "):
            cands = [m.end() if pat != r" " and not pat.startswith(" ") else m.start() + 1 for m in re.finditer(pat, seg)]
            cands = [c for c in cands if len(indent) + 24 < c <= lim and free[c - 1] and (c >= len(cur) or free[c])]
            if cands:
                best = max(cands)
                break
        if best < 0:
            break
        out.append(cur[:best].rstrip() + tail)
        cur = more + cur[best:].lstrip()
    out.append(cur + (tail if cont else ""))
    return out


def rewrap(path):
    src = open(path).read().split("\n")
    out, skipped = [], []
    skip_to = 0
    for n, line in enumerate(src, 1):
        if n <= skip_to:
            continue
        in_macro = n >= 2 and src[n - 2].rstrip().endswith("\\")
        if len(line) <= LIMIT:
            out.append(line)
            continue
        stripped = line.lstrip()
        indent = line[:len(line) - len(stripped)]
        if stripped.startswith("//"):
            m = re.match(r"(//\s*)", stripped)
            pre = m.group(1)
            body = stripped[len(pre):]
            keep = pre if len(pre) > 3 else "// "    # an aligned continuation block (`//   ...`) keeps its column
            out.extend(flow_comment(indent, body, first_prefix=pre, prefix=keep))
            continue
        cont = line.rstrip().endswith("\\")
        if stripped.startswith("#") and not cont:
            code, com = split_comment(line)
            if com is not None and len(code.rstrip()) <= LIMIT:
                out.extend(flow_comment(indent, com[2:].strip()))
                out.append(code.rstrip())
            else:
                skipped.append((n, "preprocessor"))
                out.append(line)
            continue
        code, com = split_comment(line)
        if (cont or in_macro) and com is not None:
            skipped.append((n, "macro line with a comment"))
            out.append(line)
            continue
        if com is not None:
            # the comment may go on in pure comment lines aligned under it: they move with it
            col = len(code)
            text = com[2:].strip()
            k = n
            while k < len(src) and src[k].strip().startswith("//") and abs(len(src[k]) - len(src[k].lstrip()) - col) <= 2 and col > len(indent) + 8:
                text += " " + src[k].strip()[2:].strip()
                k += 1
            skip_to = k
            out.extend(flow_comment(indent, text))
            line = code.rstrip()
            if len(line) <= LIMIT:
                out.append(line)
                continue
        if "_Pragma" in line or " asm" in line or "asm(" in line:
            skipped.append((n, "_Pragma / asm"))
            out.append(line)
            continue
        pieces = cut_code(line, cont, last_of_macro=in_macro and not cont)
        if any(len(p) > LIMIT for p in pieces):
            skipped.append((n, "no cut point"))
        out.extend(pieces)
    open(path, "w").write("\n".join(out))
    return skipped


if __name__ == "__main__":
    for p in sys.argv[1:]:
        sk = rewrap(p)
        left = sum(1 for l in open(p).read().split("\n") if len(l) > LIMIT)
        print("%-44s still > %d: %3d  %s" % (p, LIMIT, left, "; ".join("%d %s" % s for s in sk[:12])))
