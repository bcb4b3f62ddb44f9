"""CPU float32 restatement of the reference hot path (test infrastructure; see oracle/README.md)."""
