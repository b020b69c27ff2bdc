"""A declaration-level reader of Java sources -- just enough of the grammar to answer, without a JDK, the questions
a compiler answers first about the binding under integration/java: which classes, fields and methods a file declares
(name, modifiers, parameter types), what a class extends, and which members a method body uses.

Used by scripts/gen_reference_signatures.py (the reference's declarations -> tests/golden/reference_signatures.json)
and by tests/test_jni_binding.py (the binding's own declarations and uses).  Not a Java parser: comments, string and
char literals are blanked, then declarations are read at brace depth 1 of each top-level type."""
import re

_MODIFIERS = {"public", "protected", "private", "static", "final", "abstract", "native", "synchronized", "volatile",
              "transient", "strictfp", "default"}


def strip_comments_and_literals(src):
    """Comments removed, the CONTENT of string / char literals blanked (quotes kept), length and line breaks preserved."""
    out = []
    i, n = 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j]))
            i = j
        elif c in "\"'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(c + " " * (j - i - 1) + c)
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def split_top_level(text, sep=","):
    """Split on `sep` outside (), [], {}, <>."""
    parts, depth, cur = [], 0, []
    for ch in text:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur))
    return [p.strip() for p in parts if p.strip()]


def _param_names(params):
    return [re.sub(r"@\w+", " ", p).split()[-1].replace("[]", "") for p in split_top_level(params) if len(p.split()) >= 2]


def _param_types(params):
    types = []
    for p in split_top_level(params):
        toks = [t for t in re.sub(r"@\w+", " ", p).split() if t != "final"]
        if len(toks) < 2:
            continue
        name = toks[-1]
        typ = " ".join(toks[:-1])
        dims = name.count("[]")            # int name[] style
        types.append(re.sub(r"\b(?:\w+\.)+(?=\w)", "", re.sub(r"\s+", "", typ)) + "[]" * dims)   # package qualifiers dropped
    return types


def parse_types(src):
    """-> list of {name, kind, extends, implements, fields: {name: {mods, type}}, methods: [{name, mods, ret, params, body}]}
    for the top-level types of a compilation unit (nested types are skipped)."""
    clean = strip_comments_and_literals(src)
    types = []
    for m in re.finditer(r"\b(class|interface|enum)\s+(\w+)([^{;]*)\{", clean):
        # top level only: brace depth before the match must be 0
        if clean.count("{", 0, m.start()) != clean.count("}", 0, m.start()):
            continue
        head = m.group(3)
        ext = re.search(r"\bextends\s+([\w.<>, ]+?)(?=\bimplements\b|$)", head)
        imp = re.search(r"\bimplements\s+([\w.<>, ]+)$", head.strip())
        t = {"name": m.group(2), "kind": m.group(1),
             "extends": [re.sub(r"<.*>", "", e).split(".")[-1] for e in split_top_level(ext.group(1))] if ext else [],
             "implements": [re.sub(r"<.*>", "", e).split(".")[-1] for e in split_top_level(imp.group(1))] if imp else [],
             "fields": {}, "methods": []}
        # walk the body at depth 1
        i, depth, start = m.end(), 1, m.end()
        n = len(clean)
        while i < n and depth > 0:
            ch = clean[i]
            if ch == "{":
                if depth == 1:
                    decl = clean[start:i].strip()
                    # the matching close
                    j, d = i + 1, 1
                    while j < n and d > 0:
                        d += clean[j] == "{"
                        d -= clean[j] == "}"
                        j += 1
                    _take(t, decl, clean[i + 1:j - 1], m.group(1))
                    i = j
                    start = j
                    continue
                depth += 1
            elif ch == "}":
                depth -= 1
            elif ch == ";" and depth == 1:
                _take(t, clean[start:i].strip(), None, m.group(1))
                start = i + 1
            i += 1
        types.append(t)
    return types


def _take(t, decl, body, kind):
    override = "@Override" in decl
    decl = re.sub(r"@\w+(\([^)]*\))?", " ", decl).strip()
    if not decl or decl.startswith("static") and decl.strip() == "static":      # static initialiser
        return
    if "(" in decl and "=" not in decl.split("(")[0]:
        pre, rest = decl.split("(", 1)
        params = rest[:rest.rfind(")")] if ")" in rest else rest
        toks = pre.split()
        if not toks:
            return
        name = toks[-1]
        mods = [x for x in toks[:-1] if x in _MODIFIERS]
        ret = " ".join(x for x in toks[:-1] if x not in _MODIFIERS)
        if kind == "interface" and "private" not in mods and "protected" not in mods and "public" not in mods:
            mods.append("public")
        t["methods"].append({"name": name, "mods": sorted(mods), "ret": re.sub(r"\s+", "", ret),
                             "params": _param_types(params), "pnames": _param_names(params), "override": override, "body": body})
        return
    if body is not None and "=" not in decl:                                      # nested type or initialiser block
        return
    # `mods Type a = x, b[];` -> the declarators are the top-level comma parts, the first one carries the type
    parts = split_top_level(decl)
    first = parts[0].split("=")[0].split()
    mods = [x for x in first if x in _MODIFIERS]
    rest = [x for x in first if x not in _MODIFIERS]
    if len(rest) < 2:
        return
    typ = re.sub(r"\s+", "", " ".join(rest[:-1]))
    for name in [rest[-1]] + [q.split("=")[0].strip() for q in parts[1:]]:
        nm = name.replace("[]", "").strip()
        if re.fullmatch(r"\w+", nm):
            t["fields"][nm] = {"mods": sorted(mods), "type": typ + "[]" * name.count("[]")}
