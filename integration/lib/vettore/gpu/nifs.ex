defmodule Vettore.Gpu.Nifs do
  @moduledoc """
  NIF stubs for `c_src/vettore_gpu_nif.c` (libvettore_hip.so on an MI355X).

  Same conventions as `Vettore.Nifs` (lib/vettore_nifs.ex): lists of floats in,
  `{:ok, [{id, raw}]}` / `{:ok, {}}` / `{:error, binary}` out, `ArgumentError` for terms
  that do not decode.  A separate module pair instead of a replacement
  `libvettore.so`: `:erlang.load_nif/2` insists that every function of the module
  exists in the library, and 29 of Vettore.Nifs' 42 NIFs are not on the flat path.
  """
  @on_load :load

  # priv/vettore_gpu_nif.so of the application this module is compiled into (:vettore_gpu as the
  # project under integration/ builds it; :vettore if the three files are copied into Vettore
  # itself), or wherever VETTORE_GPU_NIF points (path without the extension).  libvettore_hip.so is
  # found through the NIF's rpath ($ORIGIN: the Makefile puts it beside the NIF).  A library built
  # from another include/vettore_flat.h (VT_ABI_VERSION) refuses to load: {:error, {:load_failed, _}}.
  def load do
    path =
      case System.get_env("VETTORE_GPU_NIF") do
        nil ->
          app = Application.get_application(__MODULE__) || :vettore_gpu
          :filename.join(:code.priv_dir(app), ~c"vettore_gpu_nif")

        given ->
          String.to_charlist(given)
      end

    :erlang.load_nif(path, 0)
  end

  def flat_new(_metric_code, _devices), do: :erlang.nif_error(:nif_not_loaded)
  def flat_insert(_ref, _id, _vector), do: :erlang.nif_error(:nif_not_loaded)
  def flat_insert_many(_ref, _entries), do: :erlang.nif_error(:nif_not_loaded)
  def flat_load_binary(_ref, _ids, _rows_f32_native, _dimensions), do: :erlang.nif_error(:nif_not_loaded)
  def flat_delete(_ref, _id), do: :erlang.nif_error(:nif_not_loaded)
  def flat_search(_ref, _query, _limit), do: :erlang.nif_error(:nif_not_loaded)
  def flat_search_batch(_ref, _queries, _limit), do: :erlang.nif_error(:nif_not_loaded)
  def flat_quantized_search(_ref, _query, _candidates, _limit), do: :erlang.nif_error(:nif_not_loaded)
  def flat_quantized_search_batch(_ref, _queries, _candidates, _limit), do: :erlang.nif_error(:nif_not_loaded)
  def flat_funnel_search_batch(_ref, _queries, _stages, _candidates, _limit), do: :erlang.nif_error(:nif_not_loaded)
  def flat_funnel_search(_ref, _query, _stages, _candidates, _limit), do: :erlang.nif_error(:nif_not_loaded)
  def flat_hybrid_search(_ref, _query, _generators, _limit), do: :erlang.nif_error(:nif_not_loaded)
  def normalize_l2(_vector), do: :erlang.nif_error(:nif_not_loaded)
  def compress_sign_bits(_vector), do: :erlang.nif_error(:nif_not_loaded)
  def vector_top_k(_vectors, _query, _metric_code, _dimensions, _limit), do: :erlang.nif_error(:nif_not_loaded)
  def binary_top_k(_vectors, _query, _dimensions, _limit), do: :erlang.nif_error(:nif_not_loaded)
end
