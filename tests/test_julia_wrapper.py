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
                 "jch_col_stats", "jch_plskern_fit_scaled", "jch_comm_unique_id", "jch_ctx_comm_init", "jch_ctx_comm_info"):
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
    # no n x nlv copy of T when every requested column was filled
    assert "T[:, 1:k]" not in src
