"""Poor man's pyflakes (none in this image): names a function loads that are bound nowhere it can see -- not in the
function itself (arguments, assignments, loops, withs, imports, comprehensions), not in an enclosing FUNCTION, not at
module top level, not a builtin.  The GPU-only code paths cannot be executed in the build container; this catches the
NameError class of mistakes before a GPU call is spent on them.  python tools/undefined_names.py file.py ..."""
import ast
import builtins
import sys

FUNCS = (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)


def own_bindings(scope):
    """names bound directly in `scope` (a Module, ClassDef or function): does not descend into nested functions / classes
    (their NAMES are bound here, their bodies are scopes of their own); comprehension targets count as bound here"""
    names = set()
    if isinstance(scope, FUNCS):
        a = scope.args
        for x in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
            names.add(x.arg)
    stack = list(ast.iter_child_nodes(scope))
    while stack:
        n = stack.pop()
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
            stack.extend(n.decorator_list)
            continue
        if isinstance(n, ast.Lambda):
            continue
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                names.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
        stack.extend(ast.iter_child_nodes(n))
    return names


def check(path):
    tree = ast.parse(open(path).read(), path)
    base = own_bindings(tree) | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    bad = set()

    def loads(scope):
        """Name loads directly in `scope` (not inside nested functions / lambdas / classes)"""
        stack = list(ast.iter_child_nodes(scope))
        while stack:
            n = stack.pop()
            if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)):
                stack.extend(n.decorator_list + n.args.defaults + [d for d in n.args.kw_defaults if d is not None])
                continue
            if isinstance(n, ast.Lambda):
                stack.extend(n.args.defaults + [d for d in n.args.kw_defaults if d is not None])
                continue
            if isinstance(n, ast.ClassDef):
                continue
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load):
                yield n
            stack.extend(ast.iter_child_nodes(n))

    def visit(scope, visible):
        if isinstance(scope, FUNCS):
            visible = visible | own_bindings(scope)
            for n in loads(scope):
                if n.id not in visible:
                    bad.add((n.lineno, n.id))
        elif isinstance(scope, ast.ClassDef):
            pass                                            # class-level names are not visible in its methods
        for n in ast.iter_child_nodes(scope):
            walk_nested(n, visible)

    def walk_nested(n, visible):
        if isinstance(n, FUNCS) or isinstance(n, ast.ClassDef):
            visit(n, visible)
        else:
            for c in ast.iter_child_nodes(n):
                walk_nested(c, visible)
    visit(tree, base)
    return sorted(bad)


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for line, name in check(p):
            print("%s:%d: undefined name %r" % (p, line, name))
            rc = 1
    sys.exit(rc)
