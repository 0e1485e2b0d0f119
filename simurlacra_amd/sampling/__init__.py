from .step_sequence import Step, StepSequence, discounted_reverse_cumsum, discounted_value
from .rollout import rollout
from .batch import EnvPlan, RolloutBatch, RolloutSummary, run_batch
from .parallel_sampler import ParallelRolloutSampler, ParallelSampler
from .parameter_exploration_sampler import ParameterExplorationSampler, ParameterSample, ParameterSamplingResult
from .parallel_evaluation import eval_domain_params, eval_nominal_domain, eval_randomized_domain
from .bootstrapping import bootstrap_ci, bootstrap_ci_device
from .cvar_sampler import CVaRSampler, select_cvar
