#!/usr/bin/env python3
"""Development tool (round 6 prune): resolve compile-time experiment switches of a source file to fixed values.

    python tools/resolve_switches.py file.hip NAME=VALUE ... [-U NAME ...]  > out

* `#ifndef NAME / #define NAME v / #endif` default blocks of the given names are dropped;
* `#if` / `#ifdef` / `#ifndef` / `#elif` / `#else` / `#endif` whose condition contains ONLY given names (values) or -U names (undefined)
  are evaluated and the dead branch removed; other conditionals are kept untouched;
* remaining uses of the names in code are replaced by their values (the result still compiles to the same ISA; constant expressions are then
  simplified by hand and re-checked against the ISA hash)."""
import re
import sys


def main():
    path = sys.argv[1]
    vals, undef = {}, set()
    args = sys.argv[2:]
    i = 0
    while i < len(args):
        if args[i] == '-U':
            undef.add(args[i + 1]); i += 2
        else:
            k, v = args[i].split('='); vals[k] = v; i += 1
    names = set(vals) | undef
    src = open(path).read().split('\n')
    out = []
    # stack entries: (known, taking, any_taken) - known False: conditional left as is
    stack = []

    def active():
        return all(t for k, t, _ in stack if k)

    def evaluate(cond):
        ids = set(re.findall(r'[A-Za-z_]\w*', cond)) - {'defined'}
        if not ids or not ids <= names:
            return None
        e = re.sub(r'defined\s*\(\s*(\w+)\s*\)', lambda m: '1' if m.group(1) in vals else '0', cond)
        e = re.sub(r'[A-Za-z_]\w*', lambda m: vals.get(m.group(0), '0'), e)
        e = e.replace('&&', ' and ').replace('||', ' or ').replace('!', ' not ').replace(' not =', '!=')
        return bool(eval(e))

    n = len(src)
    j = 0
    while j < n:
        line = src[j]
        st = line.strip()
        # default-definition block
        m = re.match(r'#ifndef\s+(\w+)', st)
        if m and m.group(1) in names and j + 2 < n and re.match(r'#define\s+' + m.group(1) + r'\b', src[j + 1].strip()):
            k = j + 2
            while not src[k].strip().startswith('#endif'):
                k += 1
            j = k + 1
            continue
        m = re.match(r'#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)', st)
        if m:
            kind, rest = m.group(1), m.group(2).split('//')[0].strip()
            if kind in ('if', 'ifdef', 'ifndef'):
                if kind == 'if':
                    v = evaluate(rest)
                else:
                    nm = rest.split()[0]
                    v = None if nm not in names else ((nm in vals) if kind == 'ifdef' else (nm not in vals))
                if v is None:
                    stack.append((False, True, True))
                    if active():
                        out.append(line)
                else:
                    stack.append((True, v, v))
                j += 1
                continue
            if kind == 'elif':
                k, t, a = stack[-1]
                if not k:
                    if active():
                        out.append(line)
                else:
                    v = evaluate(rest)
                    assert v is not None, 'mixed #elif: ' + line
                    stack[-1] = (True, (not a) and v, a or v)
                j += 1
                continue
            if kind == 'else':
                k, t, a = stack[-1]
                if not k:
                    if active():
                        out.append(line)
                else:
                    stack[-1] = (True, not a, True)
                j += 1
                continue
            if kind == 'endif':
                k, t, a = stack.pop()
                if not k and active():
                    out.append(line)
                j += 1
                continue
        if active():
            if not st.startswith('#'):
                code, sep, comment = line.partition('//')
                code = re.sub(r'\b(' + '|'.join(map(re.escape, vals)) + r')\b', lambda m: vals[m.group(1)], code) if vals else code
                line = code + sep + comment
            out.append(line)
        j += 1
    sys.stdout.write('\n'.join(out))


if __name__ == '__main__':
    main()
