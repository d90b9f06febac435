"""Crude undefined-name check (no pyflakes in the image): flags names a function loads that are neither bound in it or an
enclosing function, nor at module level, nor builtins.   python tools/lint_names.py file.py ..."""
import ast
import builtins
import sys


def bound_names(node):
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                out.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
    return out


def main():
    bad = 0
    for path in sys.argv[1:]:
        tree = ast.parse(open(path).read())
        known = bound_names(tree) | set(dir(builtins)) | {"__file__", "__name__"}
        for n in ast.walk(tree):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in known:
                print(f"{path}:{n.lineno}: undefined name {n.id}")
                bad += 1
    sys.exit(1 if bad else 0)


main()
