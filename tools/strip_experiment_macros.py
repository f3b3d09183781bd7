"""Remove the timing / ablation scaffolds from the product sources (round-5 review: "timing scaffolds inside product sources").

A small, purpose-built preprocessor: for a fixed set of experiment macros with their DEFAULT values it
  * drops the `#ifndef M / #define M v / #endif` blocks that declare them (and the comment lines glued to those blocks),
  * resolves every `#if` / `#else` / `#endif` whose condition mentions only those macros and integer arithmetic, keeping the live branch,
  * deletes the `F4_T(n);` stamp statements, and substitutes the remaining uses of a macro inside ordinary code with its value.
Everything else (other `#if`s, comments, layout) is left byte for byte.  The default build must be instruction-identical before and after:
tools/strip_experiment_macros.py --verify compiles both trees' device code to assembly and compares it kernel by kernel.
The reverse patch (tools/experiments/ablation_and_timing_macros.patch) restores the scaffolds for an experiment build.

usage: python tools/strip_experiment_macros.py [--check | --write | --verify OLD_TREE]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'codeformer_amd', 'csrc')
MACROS = {'F4_ABLATE': 0, 'F4_TIMING': 0, 'WS_ABLATE': 0, 'SP_ABLATE': 0, 'CF_ABLATE': 0, 'FC_ABLATE': 0, 'F4_K32_PAIR': 0, 'F4_OVL_PRIO': 0,
          'F4_RES_AUX': 0, 'WS_STORE_FIRST': 1, 'GS_CHUNK_PIN': 1,
          # second pass (same round): the remaining timing / experiment hooks
          'CF_WABLATE': 0, 'CF_W64_EXPERIMENT': 0, 'SP_DESYNC': 0, 'CF_ATTN64_GENERIC': 0}
FILES = ['cf_wf43.hip', 'cf_wsplit.hip', 'cf_split.hip', 'cf_igemm.hip', 'cf_gemm_split.hip', 'cf_winograd.hip', 'cf_attention.hip']
TOK = re.compile(r'[A-Za-z_][A-Za-z_0-9]*')


def cond_value(expr):
    """Value of a preprocessor condition if it only involves the experiment macros, else None."""
    names = TOK.findall(expr)
    if not names or any(n not in MACROS and n != 'defined' for n in names):
        return None
    e = TOK.sub(lambda m: str(MACROS[m.group(0)]), expr)
    if not re.fullmatch(r'[0-9\s()&|!=<>+\-*]+', e):
        return None
    e = e.replace('&&', ' and ').replace('||', ' or ')
    e = re.sub(r'!(?!=)', ' not ', e)
    return bool(eval(e))   # noqa: S307 (digits and operators only, checked above)


def strip(text):
    lines = text.split('\n')
    out = []
    stack = []   # entries: ('ours', live_now, taken) or ('other',)
    i = 0
    dropping = lambda: any(s[0] == 'ours' and not s[1] for s in stack)
    while i < len(lines):
        ln = lines[i]
        st = ln.strip()
        m = re.match(r'#\s*ifndef\s+(\w+)', st)
        if m and m.group(1) in MACROS and not dropping():
            # declaration block: #ifndef M [comment] / #define M v [comment] / #endif [comment]
            j = i + 1
            while j < len(lines) and not re.match(r'#\s*endif', lines[j].strip()):
                j += 1
            body = [l.strip() for l in lines[i + 1:j]]
            if all(re.match(r'#\s*define\s+' + m.group(1) + r'\b', b) or not b or b.startswith('//') for b in body):
                # drop comment lines directly above that only describe the knob
                while out and out[-1].strip().startswith('//') and m.group(1) in out[-1]:
                    out.pop()
                i = j + 1
                continue
        m = re.match(r'#\s*if\s+(.*)', st)
        if m and not st.startswith('#ifdef') and not st.startswith('#ifndef'):
            v = cond_value(m.group(1).split('//')[0].strip())
            if v is None:
                stack.append(('other',))
                if not dropping():
                    out.append(ln)
            else:
                stack.append(('ours', v, v))
            i += 1
            continue
        if re.match(r'#\s*(ifdef|ifndef)\b', st):
            stack.append(('other',))
            if not dropping():
                out.append(ln)
            i += 1
            continue
        if re.match(r'#\s*else\b', st) and stack:
            if stack[-1][0] == 'ours':
                _, live, taken = stack[-1]
                stack[-1] = ('ours', not taken, True)
            elif not dropping():
                out.append(ln)
            i += 1
            continue
        if re.match(r'#\s*elif\b', st) and stack and stack[-1][0] == 'ours':
            raise SystemExit(f'#elif on an experiment macro is not handled: {ln}')
        if re.match(r'#\s*endif\b', st) and stack:
            top = stack.pop()
            if top[0] == 'other' and not dropping():
                out.append(ln)
            i += 1
            continue
        if dropping():
            i += 1
            continue
        if re.fullmatch(r'\s*F4_T\(\d+\);\s*', ln):
            i += 1
            continue
        if st.startswith('#define F4_T('):      # the no-op definition of the stamp macro (its uses are deleted above)
            i += 1
            continue
        if not st.startswith('//'):
            code, sep, comment = ln.partition('//')
            code2 = TOK.sub(lambda mm: str(MACROS[mm.group(0)]) if mm.group(0) in MACROS else mm.group(0), code)
            ln = code2 + sep + comment
        out.append(ln)
        i += 1
    if stack:
        raise SystemExit('unbalanced #if')
    return '\n'.join(out)


def asm(path, outdir):
    os.makedirs(outdir, exist_ok=True)
    o = os.path.join(outdir, os.path.basename(path) + '.s')
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.dirname(path), '--cuda-device-only', '-S',
                    '-o', o, path], check=True)
    txt = open(o).read()
    # drop what legitimately differs: file names / line info / idents
    keep = [l for l in txt.split('\n') if not re.match(r'\s*(\.file|\.loc|\.ident|;\s*%bb|\.section\s+\.debug|\.ascii|\.string)', l) and '.hip' not in l]
    return '\n'.join(keep)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else '--check'
    if mode == '--verify':
        old = sys.argv[2]
        ok = True

        def functions(txt):
            """kernel name -> its instruction lines (labels, directives and the per-compilation __hip_cuid symbol aside)"""
            out, cur = {}, None
            for l in txt.split('\n'):
                m = re.match(r'^(_Z\w+):', l)
                if m:
                    cur = m.group(1)
                    out[cur] = []
                elif cur is not None:
                    if l.startswith('.Lfunc_end'):
                        cur = None
                    elif '__hip_cuid' not in l:
                        l = re.sub(r'\.LBB\d+_', '.LBB_', l.split(';')[0]).rstrip()    # (basic-block labels carry the function's index in the file; comments too)
                        if l:
                            out[cur].append(l)
            return out
        for f in FILES:
            ta, tb = asm(os.path.join(old, f), '/tmp/strip_old'), asm(os.path.join(CSRC, f), '/tmp/strip_new')
            # (the F(4,3) kernel lost its last template parameter with the overlapped form: <..., F32, OVL = false> is now <..., F32>)
            ta = re.sub(r'(wf43_kernelILi\d+ELi\d+ELi\d+ELi\d+ELb[01]E)Lb0E(EEvNS_6F4ArgsE)', r'\1\2', ta)
            a, b = functions(ta), functions(tb)
            gone = sorted(set(a) - set(b))
            diff = [k for k in b if a.get(k) != b[k]]
            ok &= not diff
            print(f'{f}: {len(b)} device functions, {len(b) - len(diff)} instruction-identical to the tree with the scaffolds'
                  + (f', {len(gone)} no longer instantiated' if gone else '') + (f'; DIFFER: {diff[:3]}' if diff else ''))
        sys.exit(0 if ok else 1)
    for f in FILES:
        p = os.path.join(CSRC, f)
        src = open(p).read()
        new = strip(src)
        left = [m for m in MACROS if re.search(r'\b' + m + r'\b', '\n'.join(l.split('//')[0] for l in new.split('\n')))]
        print(f'{f}: {len(src.splitlines())} -> {len(new.splitlines())} lines' + (f'; still mentions {left}' if left else ''))
        if mode == '--write':
            open(p, 'w').write(new)


if __name__ == '__main__':
    main()
