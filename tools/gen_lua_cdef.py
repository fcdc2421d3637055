"""The ffi.cdef body of lua/bot7hip_ffi.lua IS include/bot7hip.h: this prints the header with comments, the include
guard, #include / #define lines and the extern "C" wrapper stripped, one declaration per line.
tests/test_lua_shims.py compares its output with the block in the .lua file.
usage: gen_lua_cdef.py [header]          print the block and the constants
       gen_lua_cdef.py --write           rewrite both generated blocks of lua/bot7hip_ffi.lua in place
The markers inside ffi.cdef[[ ]] are C comments (the string is parsed as C, where "--" would be a syntax error)."""
import os
import re
import sys


def cdef_lines(header_text):
    t = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)           # comments
    t = re.sub(r"^\s*#.*$", "", t, flags=re.M)                        # preprocessor lines
    t = re.sub(r'extern\s+"C"\s*\{', "", t)
    t = re.sub(r"^\s*\}\s*$", "", t, flags=re.M)                      # the closing brace of extern "C"
    decls, cur, depth = [], "", 0
    for ch in t:
        cur += ch
        depth += ch == "{"
        depth -= ch == "}"
        if ch == ";" and depth == 0:
            decls.append(re.sub(r"\s+", " ", cur).strip())
            cur = ""
    return [d for d in decls if d]


def defines(header_text):
    out = {}
    for name, val in re.findall(r"^\s*#define\s+(B7_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", header_text, flags=re.M):
        out[name] = int(val)
    return out


BEGIN_CDEF = "/* BEGIN generated from include/bot7hip.h (tools/gen_lua_cdef.py) */\n"
END_CDEF = "/* END generated */"
BEGIN_CONST = "-- BEGIN generated constants\n"
END_CONST = "-- END generated constants"


def rewrite(lua_path, header_text):
    src = open(lua_path).read()
    head, rest = src.split(BEGIN_CDEF)
    _, tail = rest.split(END_CDEF, 1)
    src = head + BEGIN_CDEF + "\n".join(cdef_lines(header_text)) + "\n" + END_CDEF + tail
    head, rest = src.split(BEGIN_CONST)
    _, tail = rest.split(END_CONST, 1)
    consts = "\n".join("M.%s = %d" % (k[3:], v) for k, v in defines(header_text).items())
    open(lua_path, "w").write(head + BEGIN_CONST + consts + "\n" + END_CONST + tail)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 1 and sys.argv[1] == "--write":
        rewrite(os.path.join(root, "lua", "bot7hip_ffi.lua"), open(os.path.join(root, "include", "bot7hip.h")).read())
        sys.exit(0)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "include", "bot7hip.h")
    text = open(path).read()
    print("\n".join(cdef_lines(text)))
    print("-- constants")
    for k, v in defines(text).items():
        print("M.%s = %d" % (k[3:], v))
