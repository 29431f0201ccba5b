"""CPU: static checks of bench.py / __graft_entry__.py (they only run on the GPU box, so a slip there is not seen by the CPU suite
otherwise).  A function-level `import x` of a module that is imported at the top as well makes `x` a local of the whole function:
every use before that line raises UnboundLocalError -- on the paths the import was not written for."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shadowing_imports(path):
    tree = ast.parse(open(path).read())
    top = set()
    for node in tree.body:
        if isinstance(node, ast.Import):
            top.update(a.asname or a.name.split(".")[0] for a in node.names)
        elif isinstance(node, ast.ImportFrom):
            top.update(a.asname or a.name for a in node.names)
    bad = []
    for fn in ast.walk(tree):
        if isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)):
            for n in ast.walk(fn):
                if isinstance(n, ast.Import):
                    bad += [(fn.name, a.asname or a.name.split(".")[0]) for a in n.names if (a.asname or a.name.split(".")[0]) in top]
                elif isinstance(n, ast.ImportFrom):
                    bad += [(fn.name, a.asname or a.name) for a in n.names if (a.asname or a.name) in top]
    return bad


def test_no_function_shadows_a_module_level_import():
    for f in ("bench.py", "__graft_entry__.py"):
        assert _shadowing_imports(os.path.join(ROOT, f)) == [], f


def test_bench_compiles_and_its_argument_parser_builds():
    src = open(os.path.join(ROOT, "bench.py")).read()
    compile(src, "bench.py", "exec")
