#!/usr/bin/env python3
"""refile_gpu_tests.py - one-off (round 6, verdict housekeeping): split the GPU test files that were filed by round into files by component.
Every top-level definition keeps its text; helpers / constants / fixtures a test needs travel with it (same-named helpers of two source
files are merged when their text is equal, renamed otherwise)."""
import ast, os, re, sys
T = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')
SOURCES = ['test_gpu_parity.py', 'test_gpu_round3.py', 'test_gpu_round4.py', 'test_gpu_round5.py', 'test_gpu_dnn_bf16.py', 'test_gpu_small_calls.py']
RULES = [  # first match wins
    ('test_gpu_comm.py', ['rccl', 'clone_weights', 'a_receiver_counts']),
    ('test_gpu_dnn_bf16.py', ['bf16', 'config3']),
    ('test_gpu_small_calls.py', ['small_call', 'default_routing', '500_packet', 'fuzz_small', 'fuzz_mid', 'mid_size', 'one_packet']),
    ('test_gpu_ls.py', ['test_ls_', 'lmmse', 'fuzz_ls', 'stress_ls']),
    ('test_gpu_host_surface.py', ['csipredictor', 'keras', 'reference_model_files', 'dataset', 'test_cli', 'host_pipeline', 'estimate_c128', 'estimate_c64', 'device_resident',
                                  'hipgraph', 'empty_and_error', 'committed_oracle_fixture', 'nmse', 'profile_entry', 'engine_close', 'tensorflow_written', 'config1']),
    ('test_gpu_dnn_f32.py', ['']),
]
DOC = {
    'test_gpu_ls.py': 'LS pilot estimate (helperMIMOChannelEstimate.m:24-36 / generate_maMIMO_LTF.m:336-342) and the LMMSE smoother (LMMSE_ce.m:23-39): every LS kernel against the oracle, known channels, the reference-produced OFDM fixture',
    'test_gpu_dnn_f32.py': 'the per-pair DNN denoiser in fp32 contexts (massiveMIMO_CSI_prediction_DNN.py:176-234): fp32 MFMA kernels, split-f16 engine, fused band kernel, weight-streaming layer 0, full-size properties of BASELINE configs[1], [3], [4]',
    'test_gpu_dnn_bf16.py': 'the DNN path of bf16 contexts (BASELINE configs[2]): bf16 GEMM kernels, band kernels (8-wave and register-blocked), column split, weight-streaming layer 0',
    'test_gpu_small_calls.py': 'small and mid-size calls: the one-packet path (massiveMIMO_CSI_prediction_DNN.py:339-346), routing, fuzzed shapes, the 500-packet call of full_pipeline_maMIMO_DNNEst.sh:44-48',
    'test_gpu_host_surface.py': 'the drop-in surface: CSIPredictor / Keras-model twins (inference.py:6-68), model files, dataset / CLI, host-buffer pipeline, complex entry points, hipGraph replay, profile and metric entry points',
    'test_gpu_comm.py': 'weight transport: csi_clone_weights (the receiver side of csi_broadcast_weights), RCCL self-broadcast at world size 1',
}


def blocks(path):
    src = open(path).read()
    lines = src.split('\n')
    tree = ast.parse(src)
    out, imports = [], []
    prev_end = 0
    body = tree.body
    for i, node in enumerate(body):
        start = min([node.lineno] + [d.lineno for d in getattr(node, 'decorator_list', [])]) - 1
        # leading comment lines belong to the node
        while start > prev_end and lines[start - 1].strip().startswith('#'):
            start -= 1
        end = node.end_lineno
        text = '\n'.join(lines[start:end])
        prev_end = end
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            imports.append(text)
            continue
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant) and i == 0:
            continue                                  # module docstring
        names = []
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            names = [node.name]
        elif isinstance(node, ast.Assign):
            names = [t.id for t in node.targets if isinstance(t, ast.Name)]
        elif isinstance(node, ast.Expr):
            names = []
        used = {n.id for n in ast.walk(node) if isinstance(n, ast.Name)} | {n.value.id for n in ast.walk(node) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name)}
        # fixture arguments of tests
        if isinstance(node, ast.FunctionDef):
            used |= {a.arg for a in node.args.args}
        out.append(dict(names=names, text=text, used=used, node=node, src=os.path.basename(path)))
    return out, imports


def main():
    per_src, all_imports = {}, []
    for s in SOURCES:
        b, imp = blocks(os.path.join(T, s))
        per_src[s] = b
        all_imports += imp
    outputs = {f: [] for f, _ in RULES}
    for s in SOURCES:
        bl = per_src[s]
        defined = {}
        if s != 'test_gpu_parity.py':                 # (round 5's file imported its helpers from the parity file)
            for b in per_src['test_gpu_parity.py']:
                for n in b['names']:
                    if not n.startswith('test_'):
                        defined[n] = b
        for b in bl:
            for n in b['names']:
                defined[n] = b
        for b in bl:
            if not (b['names'] and b['names'][0].startswith('test_')):
                continue
            name = b['names'][0]
            target = next(f for f, keys in RULES if any(k in name for k in keys))
            # transitive helpers from the same source file
            need, stack = [], [b]
            seen = set()
            while stack:
                cur = stack.pop()
                for u in sorted(cur['used']):
                    d = defined.get(u)
                    if d is not None and id(d) not in seen and d is not b and not (d['names'] and d['names'][0].startswith('test_')):
                        seen.add(id(d)); need.append(d); stack.append(d)
            outputs[target].append((b, need))
    for f, items in outputs.items():
        helpers, tests, taken = [], [], {}
        for b, need in items:
            text = b['text']
            for d in sorted(need, key=lambda d: d['node'].lineno):
                for n in d['names'] or ['']:
                    if n in taken:
                        if taken[n]['text'] == d['text']:
                            continue
                        # same name, other text: rename this source's copy
                        new = '%s_%s' % (n, re.sub(r'\W', '', d['src'].replace('test_gpu_', '').replace('.py', '')))
                        key = (new,)
                        if new not in taken:
                            taken[new] = d
                            helpers.append((d, re.sub(r'\b%s\b' % re.escape(n), new, d['text'])))
                        text = re.sub(r'\b%s\b' % re.escape(n), new, text)
                    else:
                        taken[n] = d
                        helpers.append((d, d['text']))
            tests.append(text)
        # imports: keep those whose bound names are used
        body = '\n\n\n'.join([h for _, h in helpers] + tests)
        keep = []
        for imp in dict.fromkeys(all_imports):
            if 'test_gpu_parity' in imp:
                continue
            node = ast.parse(imp).body[0]
            bound = [(a.asname or a.name).split('.')[0] for a in node.names]
            if any(re.search(r'\b%s\b' % re.escape(bn), body) for bn in bound):
                keep.append(imp)
        std = sorted(i for i in keep if re.match(r'(import|from) (os|sys|json|time|subprocess|struct|ctypes|tempfile|itertools|math|re|shutil|pickle|io)\b', i))
        third = sorted(i for i in keep if i not in std and 'conftest' not in i)
        local = sorted(i for i in keep if 'conftest' in i)
        head = '"""GPU tests (-m gpu; every call through the C-ABI of libcsi_mamimo.so, checked against the numpy oracle on identical seeded inputs at the\n1e-5 norm-relative contract of BASELINE.json unless a test states its own): %s."""\n' % DOC[f]
        parts = [head, '\n'.join(std), '', '\n'.join(third), '', '\n'.join(local), '', 'pytestmark = pytest.mark.gpu']
        if re.search(r'\bsys\.path\.insert', body) is None and 'fuzz_' in body:
            parts.append('sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))')
        text = '\n'.join(parts) + '\n\n\n' + body + '\n'
        text = re.sub(r'\n{4,}', '\n\n\n', text)
        open(os.path.join(T, f + '.new'), 'w').write(text)
        print(f, len(tests), 'tests', len(helpers), 'helpers')


if __name__ == '__main__':
    main()
