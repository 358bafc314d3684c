"""CPU oracles (test infrastructure only): gssd_oracle, input_oracle, eval_oracle."""
