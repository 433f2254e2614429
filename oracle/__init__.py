"""oracle/ — TEST INFRASTRUCTURE, not product code.

A CPU restatement of the reference's (wzx99/DCFP) algorithm for the hot path, written from
the reference's semantics (every function cites the reference file:line it follows).  It
exists to CHECK the HIP path and to serve as the timed CPU baseline ("port") in bench.py.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; nothing
under dcfp_amd/ does.

Pinning: the restatement is checked against golden vectors produced by importing the real
reference in the dev container (oracle/make_golden.py -> tests/golden/*.npz), see
tests/test_oracle_vs_golden.py.  The reference has no tests or fixtures of its own
(SURVEY.md §4), so those generated vectors are the pin.
"""
