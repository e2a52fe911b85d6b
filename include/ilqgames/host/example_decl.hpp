// Declares a TopDownRenderableProblem subclass with the three Construct* hooks and the x/y/heading
// unpackers — the shape every example problem of the reference has
// (e.g. include/ilqgames/examples/modified_three_player_intersection_example.h:52-66), so the
// example translation units written for the reference compile against this mirror.
#ifndef ILQGAMES_HOST_EXAMPLE_DECL_HPP_
#define ILQGAMES_HOST_EXAMPLE_DECL_HPP_
#include <ilqgames/host/api.hpp>

// EXTRA_HOOKS: further member declarations (e.g. an overridden ConstructInitialOperatingPoint).
#define ILQGAMES_DECLARE_TOP_DOWN_EXAMPLE(ClassName) ILQGAMES_DECLARE_TOP_DOWN_EXAMPLE_WITH(ClassName, )
#define ILQGAMES_DECLARE_TOP_DOWN_EXAMPLE_WITH(ClassName, EXTRA_HOOKS) \
  namespace ilqgames {                                                \
  class ClassName : public TopDownRenderableProblem {                 \
   public:                                                            \
    ClassName() : TopDownRenderableProblem() {}                       \
    EXTRA_HOOKS                                                       \
    void ConstructDynamics() override;                                \
    void ConstructInitialState() override;                            \
    void ConstructPlayerCosts() override;                             \
    std::vector<float> Xs(const VectorXf& x) const override;          \
    std::vector<float> Ys(const VectorXf& x) const override;          \
    std::vector<float> Thetas(const VectorXf& x) const override;      \
  };                                                                  \
  }
#endif
