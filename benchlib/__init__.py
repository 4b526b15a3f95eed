"""Parts of bench.py: config (sizes, peaks), synth (inputs), cpu_legs (oracle timings), roofline (work models), record
(the compact line), workloads (the secondary BASELINE configs)."""
