"""The measuring legs of bench.py (one module per child-process role).  bench.py itself is the argument parser, the
orchestrator that never touches the GPU, and the process plumbing."""
