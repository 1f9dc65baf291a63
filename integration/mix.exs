defmodule VettoreGpu.MixProject do
  use Mix.Project

  # `Vettore.Index.FlatGpu` + `Vettore.Gpu.Nifs` as a package of their own beside :vettore (the
  # plugin point is `Vettore.new(index: SomeModule, index_options: ...)`, lib/vettore/index.ex:12-17,
  # collection.ex:72, :1283-1298 -- nothing in :vettore changes).  One command on a box with OTP,
  # ROCm and an MI355X:
  #
  #     VETTORE_HIP_ROOT=/path/to/this/repository mix test
  #
  # `mix compile` runs the Makefile beside this file (elixir_make): it builds
  # priv/vettore_gpu_nif.so from c_src/vettore_gpu_nif.c against $(ERTS_INCLUDE_DIR), links
  # libvettore_hip.so (built by `make` at the repository root) and copies it into priv/ so that the
  # NIF's rpath ($ORIGIN) finds it after a release is assembled.
  def project do
    [
      app: :vettore_gpu,
      version: "0.1.0",
      elixir: "~> 1.18",
      compilers: [:elixir_make] ++ Mix.compilers(),
      make_targets: ["all"],
      make_clean: ["clean"],
      make_env: fn ->
        %{"VETTORE_HIP_ROOT" => System.get_env("VETTORE_HIP_ROOT") || Path.expand("..", __DIR__)}
      end,
      start_permanent: Mix.env() == :prod,
      deps: deps()
    ]
  end

  def application, do: [extra_applications: [:logger]]

  defp deps do
    [
      # the collection, the ETS store and the `Vettore.Index` behaviour (checked out next to this
      # repository: `VETTORE_PATH=../../vettore mix deps.get`; else the hex package)
      vettore_dep(),
      {:elixir_make, "~> 0.8", runtime: false}
    ]
  end

  defp vettore_dep do
    case System.get_env("VETTORE_PATH") do
      nil -> {:vettore, "~> 0.3.2"}
      path -> {:vettore, path: path}
    end
  end
end
