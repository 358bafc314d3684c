"""MI355X engine of the GSSD detection path: C-ABI binding, launch plans, input stage, evaluator."""
