# which test of a file leaves an unread HIP error behind (torch reports it at its next call): run each test, then allocate through torch in the same process
cd $GRAFT_REPO_ROOT
for t in $(python -m pytest tests/test_abi_extras.py -m gpu --collect-only -q 2>/dev/null | grep "::"); do
  python - "$t" <<'PY' 2>&1 | tail -1
import sys, pytest, torch
t = sys.argv[1]
rc = pytest.main(["-q", "-x", "-m", "gpu", t, "-p", "no:cacheprovider"])
try:
    torch.empty(1024, device="cuda"); torch.cuda.synchronize()
    print(t, "rc", rc, "clean")
except Exception as e:
    print(t, "rc", rc, "STICKY", str(e).splitlines()[0])
PY
done
