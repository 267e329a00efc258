"""The multi-GPU exchange logic of rt_reduce_accum / rt_reduce_scatter_rows (csrc/rt_reduce.cpp) driven through an
injected fake RCCL table (rt_debug_reduce_with_fake: only in librt_reduce_test.so, the -DRT_TEST_HOOKS build of that file): no machine available to the build has two GPUs, so the argument checks, the communicator
cache and the exact sequence of grouped ncclReduce calls are pinned here, on the CPU, call by call."""
import ctypes as C


def run(lib, mode, devices, count=0, root=0, w=0, h=0, null_buffer=-1, fail_at=-1, reset=False):
    fn = lib.rt_debug_reduce_with_fake
    fn.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                   C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    n = len(devices)
    dev = (C.c_int * max(1, n))(*devices)
    rows = (C.c_int * (n + 2))()
    log = C.create_string_buffer(1 << 16)
    rc = fn(mode, dev, n, count, root, w, h, rows, null_buffer, fail_at, 1 if reset else 0, log, len(log))
    return rc, log.value.decode().splitlines(), list(rows)[:n + 1]


def test_reduce_accum_call_sequence_and_communicator_cache(binding):
    lib = binding.load_reduce_test_library()
    rc, log, _ = run(lib, 0, [0, 1, 2, 3], count=3 * 64 * 48, root=0, reset=True)
    assert rc == 0
    assert log[0] == "init dev0=comm1 dev1=comm2 dev2=comm3 dev3=comm4"
    assert log[1] == "group_start" and log[-1] == "group_end" and len(log) == 7
    for i, line in enumerate(log[2:6]):
        # rank i: its own device made current, its own communicator, buffer and stream; in place, float sum, root 0
        assert line == "reduce dev%d comm%d rank_buf%d off0 count%d root0 stream%d inplace f32sum" % (i, i + 1, i, 3 * 64 * 48, 100 + i)
    # same device list again: the communicators are reused, nothing is initialised or destroyed
    rc, log, _ = run(lib, 0, [0, 1, 2, 3], count=10, root=2)
    assert rc == 0 and not any(s.startswith(("init", "destroy")) for s in log)
    assert all(" root2 " in s for s in log if s.startswith("reduce")) and sum(s.startswith("reduce") for s in log) == 4
    # another list (order matters: rank = position): the old ones are destroyed, new ones made
    rc, log, _ = run(lib, 0, [1, 0], count=10, root=1)
    assert rc == 0
    assert log[:5] == ["destroy comm1", "destroy comm2", "destroy comm3", "destroy comm4", "init dev1=comm5 dev0=comm6"]
    assert log[6] == "reduce dev1 comm5 rank_buf0 off0 count10 root1 stream100 inplace f32sum"
    assert log[7] == "reduce dev0 comm6 rank_buf1 off0 count10 root1 stream101 inplace f32sum"
    # one device: still a (trivial) reduce, so that callers need no special case
    rc, log, _ = run(lib, 0, [5], count=7, root=0)
    assert rc == 0 and sum(s.startswith("reduce") for s in log) == 1


def test_reduce_accum_refuses_bad_arguments_before_touching_rccl(binding):
    lib = binding.load_reduce_test_library()
    for kw in (dict(devices=[0, 1], count=0, root=0), dict(devices=[0, 1], count=8, root=2), dict(devices=[0, 1], count=8, root=-1),
               dict(devices=[], count=8, root=0), dict(devices=[0, 0], count=8, root=0), dict(devices=[0, 1, 0], count=8, root=0),
               dict(devices=[0, 1], count=8, root=0, null_buffer=1)):
        devices = kw.pop("devices")
        rc, log, _ = run(lib, 0, devices, reset=True, **kw)
        assert rc == binding_err_arg(), (devices, kw)
        assert log == [], "RCCL was called for a refused request"
    assert b"two contexts on one device" in lib.rt_last_error() or b"null context" in lib.rt_last_error()


def binding_err_arg():
    return -1       # RT_ERR_ARG (include/rt_api.h)


def test_a_failing_reduce_closes_the_group_and_reports(binding):
    lib = binding.load_reduce_test_library()
    rc, log, _ = run(lib, 0, [0, 1, 2], count=5, root=0, fail_at=1, reset=True)
    assert rc == -5                                                     # RT_ERR_RCCL
    assert log[-1] == "group_end" and sum(s.startswith("reduce") for s in log) == 2     # stopped at the failure, group closed
    assert b"fake failure" in lib.rt_last_error()
    # the communicators survived the failure and serve the next call
    rc, log, _ = run(lib, 0, [0, 1, 2], count=5, root=0)
    assert rc == 0 and not any(s.startswith("init") for s in log)


def test_reduce_scatter_rows_bands(binding):
    lib = binding.load_reduce_test_library()
    W, H = 7, 10
    rc, log, rows = run(lib, 1, [0, 1, 2], w=W, h=H, reset=True)
    assert rc == 0 and rows == [0, 3, 6, 10]                            # bands may differ in height
    reduces = [s for s in log if s.startswith("reduce")]
    assert len(reduces) == 9 and log[1] == "group_start" and log[-1] == "group_end"     # ONE group for all bands
    k = 0
    for band in range(3):
        off, cnt = rows[band] * W * 3, (rows[band + 1] - rows[band]) * W * 3
        for i in range(3):
            assert reduces[k] == "reduce dev%d comm%d rank_buf%d off%d count%d root%d stream%d inplace f32sum" % (i, i + 1, i, off, cnt, band, 100 + i)
            k += 1
    # more GPUs than rows: empty bands are skipped, not sent as zero-length reduces
    rc, log, rows = run(lib, 1, [0, 1, 2, 3], w=5, h=2, reset=True)
    assert rc == 0 and rows == [0, 0, 1, 1, 2]
    assert sum(s.startswith("reduce") for s in log) == 8 and all(" count15 " in s for s in log if s.startswith("reduce"))
    rc, log, _ = run(lib, 1, [0, 1], w=0, h=4, reset=True)
    assert rc == -1 and log == []
