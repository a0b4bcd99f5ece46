"""myfm_amd -- MI355X-native Gibbs sampler for Bayesian Factorization Machines (drop-in for the
MyFMRegressor / MyFMClassifier / RelationBlock path of tohtsky/myFM)."""
