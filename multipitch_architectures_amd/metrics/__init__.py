from .eval_metrics import calculate_single_measure, calculate_eval_measures, MEASURES  # noqa: F401
