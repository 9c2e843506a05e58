"""CPU restatement of the reference algorithms; test infrastructure only."""
