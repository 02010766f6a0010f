"""Readable kernel names for the profile summaries.

rocprofv3 7.2 leaves names whose template arguments include `_Float16` / `__bf16` (Itanium `DF16_` / `DF16b`) mangled,
and neither binutils' c++filt nor anything else in the image demangles them.  `pretty()` handles the subset this
library's kernels use (anonymous-namespace function templates over types, ints and bools) and spells the result the
way `vidil_gemm_kernel_name()` does, so a profile row and a bench.py `roofline.kernel` are the same string.
"""
import re

_BUILTIN = {"f": "float", "i": "int", "j": "unsigned int", "h": "unsigned char", "a": "signed char", "s": "short",
            "t": "unsigned short", "l": "long", "m": "unsigned long", "b": "bool", "d": "double", "c": "char", "v": "void"}


def _template_args(s):
    """Parse `I…E` starting at s[0] == 'I'; returns (list of spelled arguments, rest) or None when a token is unknown."""
    out, i = [], 1
    while i < len(s) and s[i] != "E":
        if s.startswith("DF16_", i):
            out.append("_Float16"); i += 5
        elif s.startswith("DF16b", i):
            out.append("__bf16"); i += 5
        elif (m := re.match(r"L([ijlmstah])(n?\d+)E", s[i:])):
            out.append(m.group(2).replace("n", "-")); i += m.end()
        elif (m := re.match(r"Lb([01])E", s[i:])):
            out.append("true" if m.group(1) == "1" else "false"); i += m.end()
        elif (m := re.match(r"N(?:S_|12_GLOBAL__N_1)(\d+)", s[i:])):
            n = int(m.group(1)); j = i + m.end()
            out.append(s[j:j + n]); i = j + n
            if s[i:i + 1] != "E":
                return None
            i += 1
        elif (m := re.match(r"(\d+)", s[i:])):                       # a global struct (fp8)
            n = int(m.group(1)); j = i + m.end()
            out.append(s[j:j + n]); i = j + n
        elif s[i] in _BUILTIN:
            out.append(_BUILTIN[s[i]]); i += 1
        else:
            return None
    return out, s[i + 1:]


def pretty(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:
        return re.sub(r"\(.*$", "", name)
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if rest.startswith("I"):
        parsed = _template_args(rest)
        if parsed is None:
            return name
        return f"{base}<{', '.join(parsed[0])}>"
    return base


if __name__ == "__main__":
    import sys
    for line in sys.stdin:
        print(pretty(line.strip()))
