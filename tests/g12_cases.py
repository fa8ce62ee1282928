"""Golden set G12 (tests/golden/make_golden_g12.py): center() / standardize() on column-major, strided column-major and
single-column count matrices of five dtypes, as the reference leaves them.  One comparison for two runners: the oracle's
restatement (tests/test_oracle_golden.py, CPU) and seekr_amd.BasicCounter (tests/test_gpu_parity.py, on the GPU)."""
import json
import os

import make_golden_g12 as mk


def check_all(golden_dir, counter_cls):
    with open(os.path.join(golden_dir, "g12_column_major.json")) as fh:
        want = json.load(fh)["cases"]
    n = 0
    for dtype in mk.DTYPES:
        for shape in mk.SHAPES:
            for layout in mk.LAYOUTS:
                key = "%s_%dx%d_%s" % (dtype, shape[0], shape[1], layout)
                got = mk.run(counter_cls, dtype, shape, layout)
                assert got == want[key], (key, got, want[key])
                n += 1
    return n


def oracle_counter(orc):
    """BasicCounter's three attributes and two methods over the oracle's restatement (host_center / host_standardize)."""
    class OracleCounter:
        def __init__(self, silent=True, k=1):
            self.mean, self.std, self.counts = True, True, None

        def center(self):
            self.mean, op = orc.host_center(self.counts, self.mean)  # the attribute is replaced before the operation
            op()

        def standardize(self):
            self.std, op = orc.host_standardize(self.counts, self.std)
            op()
    return OracleCounter
