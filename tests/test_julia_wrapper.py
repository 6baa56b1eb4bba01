"""Static checks of the Julia host layer (jchemo.jl_amd/julia/JchemoHIP.jl) against the C ABI header.

There is no Julia toolchain in the build image, so the wrapper cannot be executed here; these tests make it
correct by construction where a machine can check it:
  * every `ccall` spells its argument-type tuple out literally (a variable there is a lowering error in Julia —
    the round-1 defect), and passes exactly as many arguments as the tuple has types;
  * for every `jch_*` entry point the tuple has the arity of the prototype in include/jchemo_hip.h and each
    position has the matching kind (Int32 / Int64 / UInt32 / Float64 / pointer), and the return type matches;
  * the fit wrappers build `Jchemo.Plsr` (reference record, src/plskern.jl:1-14) with the fields in the reference order.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "jchemo.jl_amd", "julia", "JchemoHIP.jl")
HDR = os.path.join(ROOT, "include", "jchemo_hip.h")


def _balanced(src, i):
    """src[i] == '(' -> index one past its matching ')' (string literals and comments are not expected inside)."""
    depth = 0
    for j in range(i, len(src)):
        c = src[j]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return j + 1
    raise AssertionError("unbalanced parenthesis")


def _split_top(s):
    out, depth, cur = [], 0, []
    for c in s:
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        if c == "," and depth == 0:
            out.append("".join(cur).strip())
            cur = []
        else:
            cur.append(c)
    tail = "".join(cur).strip()
    if tail:
        out.append(tail)
    return out



def julia_ccalls():
    src = open(JL).read()
    code = "\n".join(re.sub(r"#(?![^\"]*\"[^\"]*$).*$", "", ln) for ln in src.splitlines())
    calls = []
    for m in re.finditer(r"\bccall\(", code):
        end = _balanced(code, m.end() - 1)
        args = _split_top(code[m.end():end - 1])
        calls.append(args)
    return calls


def header_protos():
    h = open(HDR).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    protos = {}
    for m in re.finditer(r"JCH_API\s+([\w\s\*]+?)\b(jch_\w+)\s*\(([^;]*?)\)\s*;", h, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        plist = [] if params.strip() == "void" else [re.sub(r"\s+", " ", p.strip()) for p in params.split(",")]
        protos[name] = (ret, plist)
    return protos


def c_kind(param):
    if "*" in param:
        return "ptr"
    t = param.replace("const ", "").split()[0]
    return {"int32_t": "Int32", "int64_t": "Int64", "uint32_t": "UInt32", "uint64_t": "UInt64", "double": "Float64"}[t]


def jl_kind(t):
    t = t.strip()
    if t.startswith("Ptr{") or t.startswith("Ref{") or t == "Cstring":
        return "ptr"
    return t


def test_every_ccall_has_a_literal_signature_and_matching_argument_count():
    calls = julia_ccalls()
    assert len(calls) >= 15
    for args in calls:
        assert len(args) >= 3, args
        sig = args[2]
        assert sig.startswith("(") and sig.endswith(")"), f"ccall signature is not a literal tuple: {sig!r} in {args[0]}"
        types = _split_top(sig[1:-1])
        for t in types:
            assert re.fullmatch(r"[A-Za-z_][\w]*(\{.*\})?", t), f"not a literal type: {t!r}"
        assert len(args) - 3 == len(types), f"{args[0]}: {len(types)} types, {len(args) - 3} arguments"


def test_ccall_signatures_match_the_header():
    protos = header_protos()
    assert "jch_plskern_fit" in protos and "jch_lwplsr_predict" in protos
    seen = set()
    for args in julia_ccalls():
        target = args[0]
        m = re.match(r"\(:(\w+),\s*LIB\)", target)
        names = [m.group(1)] if m else None
        if names is None and target == "($cname, LIB)":      # the @eval-generated fit wrappers
            names = ["jch_plskern_fit", "jch_plsnipals_fit", "jch_plssimp_fit", "jch_plsrosa_fit"]
        if names is None:
            assert "jl_generating_output" in target, f"unexpected ccall target {target!r}"
            continue
        types = _split_top(args[2][1:-1])
        for name in names:
            assert name in protos, f"{name} is not declared in include/jchemo_hip.h"
            ret, params = protos[name]
            assert len(types) == len(params), f"{name}: header has {len(params)} parameters, the ccall {len(types)}"
            for pos, (jt, cp) in enumerate(zip(types, params)):
                assert jl_kind(jt) == c_kind(cp), f"{name} argument {pos}: Julia {jt} vs C '{cp}'"
            want_ret = "Cstring" if "char" in ret else {"int32_t": "Int32"}[ret]
            assert args[1].strip() == want_ret, f"{name}: return type {args[1]} vs C {ret}"
            seen.add(name)
    # the drop-in surface of the north star and every §8 entry point the wrapper claims
    for name in ("jch_ctx_create", "jch_ctx_destroy", "jch_last_error", "jch_plskern_fit", "jch_plsnipals_fit", "jch_plssimp_fit",
                 "jch_plsrosa_fit", "jch_plswold_fit", "jch_affine_gemm", "jch_weighted_ss", "jch_lwplsr_predict", "jch_weighted_cov",
                 "jch_col_stats", "jch_plskern_fit_scaled", "jch_comm_unique_id", "jch_ctx_comm_init", "jch_ctx_comm_info",
                 "jch_score_sums", "jch_lwplsr_prepare", "jch_lwplsr_predict_prepared", "jch_lwplsr_release"):
        assert name in seen, f"the Julia wrapper never calls {name}"


def test_pls_desc_mirrors_the_c_struct():
    h = open(HDR).read()
    body = re.search(r"typedef struct jch_pls_desc \{(.*?)\} jch_pls_desc;", h, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    cfields = [(t, n) for t, n in re.findall(r"(int64_t|int32_t)\s+(\w+)\s*;", body)]
    src = open(JL).read()
    jbody = re.search(r"struct PlsDesc[^\n]*\n(.*?)\nend", src, flags=re.S).group(1)
    jfields = re.findall(r"(\w+)::(Int64|Int32)", jbody)
    assert [(n, {"int64_t": "Int64", "int32_t": "Int32"}[t]) for t, n in cfields] == jfields


def test_fit_returns_the_reference_record_in_reference_field_order():
    src = open(JL).read()
    # Jchemo.Plsr(T, P, R, W, C, TT, xmeans, xscales, ymeans, yscales, weights, niter)   (src/plskern.jl:1-14)
    assert re.search(r"getfield\(J, :Plsr\), T, P, R, W, C, TT, xm, xs, ym, ys, wn, niter\)", src)
    fields = re.search(r"struct Plsr\{TT_, WT\}(.*?)\nend", src, flags=re.S).group(1)
    names = re.findall(r"^\s+(\w+)::", fields, flags=re.M)
    assert names == ["T", "P", "R", "W", "C", "TT", "xmeans", "xscales", "ymeans", "yscales", "weights", "niter"]
    assert "Base.summary(object::Plsr, X" in src and "explvarx" in src
    assert "fbca9394-dd0a-4d1c-b066-ae75f6ef1ad5" in src      # Jchemo's package uuid (reference Project.toml:2)
    # no n x nlv copy of T when every requested column was filled (the fit's `cut` helper; `object.T[:, 1:k]` of vip is a model slice)
    assert not re.search(r"(?<![.\w])T\[:, 1:k\]", src)


# ---- round 3: the §8(f) surface in Julia, and structural checks a machine can do without a Julia toolchain ------------------
def _strip_strings_and_comments(src):
    """Julia source with string literals (incl. triple-quoted docstrings) blanked and `#` comments removed; same length."""
    out, i, n = [], 0, len(src)
    while i < n:
        if src.startswith('"""', i):
            j = src.index('"""', i + 3) + 3
            out.append("".join(c if c == "\n" else " " for c in src[i:j])); i = j
        elif src[i] == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('"' + " " * (j - i - 1) + '"'); i = j + 1
        elif src[i] == "#":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src[i] == "'" and i + 2 < n and src[i + 2] == "'":      # character literal 'x'
            out.append("   "); i += 3
        else:
            out.append(src[i]); i += 1
    return "".join(out)


def test_julia_blocks_and_brackets_balance_per_top_level_form():
    """Every top-level form closes what it opens: (), [], {} and the block keywords against `end` (an `end` inside [...] is
    an index, `for` / `if` inside brackets are generators).  A missing `end` or parenthesis is the commonest way an
    unexecuted file is wrong."""
    code = _strip_strings_and_comments(open(JL).read())
    assert len(code) == len(open(JL).read())
    openers = {"function", "if", "for", "while", "begin", "let", "struct", "module", "do", "try", "quote", "macro"}
    depth_blk, stack = 0, []
    tok = re.compile(r"[A-Za-z_!][\w!]*|[()\[\]{}]")
    prev = ""
    line_start_depth = {}
    for m in tok.finditer(code):
        t = m.group(0)
        if t in "([{":
            stack.append((t, m.start()))
        elif t in ")]}":
            assert stack, f"unmatched {t} at offset {m.start()}: ...{code[max(0, m.start() - 60):m.start() + 1]!r}"
            o, _ = stack.pop()
            assert {"(": ")", "[": "]", "{": "}"}[o] == t, f"{o} closed by {t} near {code[max(0, m.start() - 60):m.start() + 1]!r}"
        elif not stack or all(o == "(" for o, _ in stack) and False:
            pass
        if t.isidentifier() or t.endswith("!"):
            inside_brackets = any(o in "[{" for o, _ in stack)
            inside_parens = bool(stack)
            if t in openers and not inside_parens:
                if t == "struct" and prev == "mutable":
                    depth_blk += 1
                elif t == "if" and prev == "else":      # `else if` does not exist in Julia; elseif is one token
                    depth_blk += 1
                else:
                    depth_blk += 1
            elif t == "end" and not inside_brackets:
                if not inside_parens:
                    depth_blk -= 1
                    assert depth_blk >= 0, f"`end` without an opener near {code[max(0, m.start() - 80):m.start() + 3]!r}"
            prev = t
    assert not stack, f"unclosed {stack[-1][0]} opened at ...{code[stack[-1][1]:stack[-1][1] + 80]!r}"
    assert depth_blk == 0, f"{depth_blk} block(s) left open (module ... end included)"


def test_julia_exports_cover_the_python_mirror():
    """Every name of the executed host mirror (jchemo_hip/__init__.py) that has a reference counterpart is exported by the
    Julia module too (gridscorelv / gridcvlv / plsrda / mbplsr / vip / xfit existed only in Python)."""
    src = open(JL).read()
    exported = set(re.findall(r"[\w!]+", re.search(r"\nexport (.*?)\n\n", src, flags=re.S).group(1)))
    must = ["plskern", "plskern!", "plsnipals", "plsnipals!", "plssimp", "plssimp!", "plsrosa", "plsrosa!", "plswold", "plswold!", "transform", "coef",
            "predict", "lwplsr", "msep", "rmsep", "ssr", "bias", "r2", "cor2", "mpar", "segmkf", "segmts", "gridscorelv", "gridcvlv", "dummy",
            "plsrda", "mbplsr", "vip", "xfit", "xresid", "Plsr", "Lwplsr", "Plsrda", "Mbplsr"]
    missing = [nm for nm in must if nm not in exported]
    assert not missing, missing
    init = open(os.path.join(ROOT, "jchemo.jl_amd", "jchemo_hip", "__init__.py")).read()
    for nm in ("gridscorelv", "gridcvlv", "plsrda", "mbplsr", "vip", "xfit", "xresid", "mpar", "segmkf", "segmts"):
        assert nm in init                                       # ... and they are the mirror's names
    for nm in must:
        if nm[0].islower():
            assert re.search(r"(^|\n)\s*(function\s+)?(Base\.)?" + re.escape(nm) + r"\(", src) or re.search(r"const " + re.escape(nm) + r"\b", src), f"{nm} exported but never defined"


def test_julia_keyword_names_equal_the_reference_signatures():
    """Keyword names of the reference's call shapes (file:line of /root/reference/src cited), which higher-order callers pass by
    name: the Julia mirror must accept exactly these (plus `ctx`)."""
    src = open(JL).read()
    want = {
        "gridscorelv": (["Xtrain", "Ytrain", "X", "Y"], ["score", "fun", "nlv", "pars", "verbose"]),       # src/gridscore.jl:167-168
        "gridcvlv": (["X", "Y"], ["segm", "score", "fun", "nlv", "pars", "verbose"]),                      # src/gridcv.jl:187-188
        "plsrda": (["X", "y", "weights"], ["nlv", "scal"]),                                                # src/plsrda.jl:71-72
        "mbplsr": (["Xbl", "Y", "weights"], ["nlv", "bscal", "scal"]),                                     # src/mbplsr.jl:64-65
        "lwplsr": (["X", "Y"], ["nlvdis", "metric", "h", "k", "nlv", "tol", "scal", "verbose"]),           # src/lwplsr.jl:114-115
        "xfit": (["object", "X"], ["nlv"]), "xresid": (["object", "X"], ["nlv"]),                          # src/xfit.jl:37,86
        "plswold": (["X", "Y", "weights"], ["nlv", "tol", "maxit", "scal"]),                               # src/plswold.jl:30-31
    }
    for fn, (pos, kws) in want.items():
        m = re.search(r"(?:^|\n)(?:function )?" + fn + r"\(", src)
        assert m, f"no definition found for {fn}"
        arglist = src[m.end():_balanced(src, m.end() - 1) - 1]
        assert ";" in arglist, f"{fn} has no keyword section"
        head, tail = arglist.split(";", 1)
        got_pos = [re.split(r"[:=\s]", a.strip())[0] for a in _split_top(head) if a.strip()]
        got_kw = [re.split(r"[:=\s]", a.strip())[0] for a in _split_top(tail) if a.strip()]
        assert got_pos == pos, (fn, got_pos)
        assert [k_ for k_ in got_kw if k_ != "ctx"] == kws, (fn, got_kw)


def test_every_pointer_handed_to_a_ccall_is_gc_preserved():
    """`pointer(A)` of a Julia array is only valid while A is rooted: every ccall that takes `pointer(name)` must sit inside a
    `GC.@preserve ... name ...` expression (a closure or a later statement would not keep the array alive)."""
    code = _strip_strings_and_comments(open(JL).read())
    for m in re.finditer(r"\bccall\(", code):
        end = _balanced(code, m.end() - 1)
        body = code[m.end():end]
        names = set(re.findall(r"\bpointer\((\w+)\)", body))
        if not names:
            continue
        k = code.rfind("GC.@preserve", 0, m.start())
        assert k >= 0 and m.start() - k < 1500, f"ccall with pointer({sorted(names)}) outside any GC.@preserve: {body[:80]!r}"
        header = code[k + len("GC.@preserve"):m.start()]
        kept = set(re.findall(r"\w+", header.split("begin")[0].split("check(")[0].split("\n")[0]))
        missing = names - kept
        assert not missing, f"pointer({sorted(missing)}) is not listed in the enclosing GC.@preserve ({sorted(kept)})"


REF = "/root/reference/src"


def _fn_body(src, name):
    """Text of `function name(...) ... end` (first definition), found by block balance (keywords inside (), [] are generators /
    indices, as in the balance test above)."""
    m = re.search(r"(?:^|\n)function " + re.escape(name) + r"\(", src)
    assert m, f"no `function {name}(`"
    start = m.start() + (1 if src[m.start()] == "\n" else 0)
    code = _strip_strings_and_comments(src[start:])
    openers = {"function", "if", "for", "while", "begin", "let", "struct", "do", "try", "quote"}
    depth, nest = 0, 0
    for t in re.finditer(r"[A-Za-z_!][\w!]*|[()\[\]{}]", code):
        w = t.group(0)
        if w in "([{":
            nest += 1
        elif w in ")]}":
            nest -= 1
        elif nest == 0 and w in openers:
            depth += 1
        elif nest == 0 and w == "end":
            depth -= 1
            if depth == 0:
                return src[start:start + t.end()]
    raise AssertionError(f"unterminated function {name}")


def test_julia_grid_functions_return_the_reference_tables():
    """Round-3 review item 8: `gridscorelv` / `gridcvlv` / `explvarx` return what the reference returns — DataFrames (when that
    package is loaded; it is a dependency of Jchemo) with the reference's column names in the reference's order.  The column names
    are read from the reference SOURCE when it is present (this container); the Julia side is checked statically (NOT EXECUTED: no
    Julia toolchain here)."""
    src = open(JL).read()
    # the tables go through ONE constructor that yields a DataFrame when DataFrames is loaded
    assert "a93c6f00-e57d-5684-b7b6-d8193f3e46c0" in src and "dataframes_module()" in src
    tbl = _fn_body(src, "_table")
    assert ":DataFrame" in tbl and "invokelatest" in tbl
    # expected column names: from the reference source where available
    want_rep, want_expl, want_y = ["repl", "segm"], ["nlv", "var", "pvar", "cumpvar"], 'Symbol("y", i)'
    if os.path.isdir(REF):
        gcv = open(os.path.join(REF, "gridcv.jl")).read()
        body = gcv[gcv.index("function gridcvlv("):]
        body = body[:body.index("\nend\n")]
        m = re.search(r"DataFrame\(repl = .*?\n\s*segm = ", body, flags=re.S)
        assert m, "reference gridcvlv no longer builds DataFrame(repl = ..., segm = ...)"
        assert re.search(r"\(res = res, res_rep = res_rep,? *\)", body)                      # src/gridcv.jl:227
        assert "namgroup = [:nlv]" in body and "[:nlv ; collect(keys(pars))]" in body          # :223
        gsc = open(os.path.join(REF, "gridscore.jl")).read()
        gb = gsc[gsc.index("function gridscorelv("):]
        gb = gb[:gb.index("\nend\n")]
        assert "hcat(dat, DataFrame(nlv = znlv))" in gb and "hcat(dat, res)" in gb             # pars columns, then nlv, then y1 ... yq
        assert 'repeat(["y"], q), 1:q' in gb
        pk = open(os.path.join(REF, "plskern.jl")).read()
        m = re.search(r"explvarx = DataFrame\((.*?)\)", pk)
        assert m and [a.split("=")[0].strip() for a in m.group(1).split(",")] == want_expl   # src/plskern.jl:258
    # explvarx: the four columns in that order, through _table
    ex = _fn_body(src, "explvarx")
    m = re.search(r"_table\(\((.*?)\)\)", ex, flags=re.S)
    assert m and [a.split("=")[0].strip() for a in _split_top(m.group(1))] == want_expl
    # _grid_cols: pars columns first, then nlv, then y1 ... yq
    gc_ = _fn_body(src, "_grid_cols")
    i_pars, i_nlv, i_y = gc_.index("keys(pars)"), gc_.index("(nlv = "), gc_.index("_ynames(")
    assert i_pars < i_nlv < i_y
    assert want_y in src[src.index("_ynames(q) ="):src.index("_ynames(q) =") + 120]
    # gridscorelv returns the table itself; gridcvlv the pair (res, res_rep) with repl, segm in front and the group keys of res = nlv, pars...
    gs = _fn_body(src, "gridscorelv")
    assert re.search(r"\n\s*_table\(_grid_cols\(pars, rng, .*\)\)\nend$", gs)
    gv = _fn_body(src, "gridcvlv")
    assert re.search(r"\(res = _table\(res_cols\), res_rep = _table\(rep_cols\)\)\nend$", gv)
    m = re.search(r"rep_cols = merge\(\((.*?)\), stacked\)", gv)
    assert m and [a.split("=")[0].strip() for a in _split_top(m.group(1))] == want_rep
    assert "(:nlv, keys(pars)...)" in gv and "(:nlv,)" in gv
    assert "NOT EXECUTED" in src and "NOT executed" in open(os.path.join(ROOT, "INTEGRATION.md")).read()   # the banners stay
